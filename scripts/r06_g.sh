#!/bin/bash
mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline --no-catalogue --no-extra --steps 30 --warmup 5"
for rep in 1 2; do for m in x6 x6l1 fp32; do CLSR_ATT_BWD=$m $B 2> /dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"])' $m; done; done
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "compaction" 2>&1 | tail -2
python bench.py --no-cpu-baseline --config catalogue100m --no-extra --steps 8 2> gpurun_out/r06g_cat.err | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("catalogue", d["ms_per_step"])'
bash scripts/prof_step.sh r06g_cat --config catalogue100m --steps 8 | tail -1

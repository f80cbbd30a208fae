#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_att_bwd_x3_gpu.py tests/test_switches_gpu.py -x -q -m gpu -k "layer0_backward or switch" > gpurun_out/r06j_tests.log 2>&1; echo "tests rc $?"; tail -2 gpurun_out/r06j_tests.log
B="python bench.py --no-cpu-baseline --no-catalogue --no-extra --steps 30 --warmup 5"
for rep in 1 2 3; do for m in x6 x6l1; do CLSR_ATT_BWD=$m $B 2> /dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"])' $m; done; done
CLSR_ATT_BWD=x6 bash scripts/prof_step.sh r06j_x6 | tail -1

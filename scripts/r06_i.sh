#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_x3_gpu.py tests/test_step_gpu.py -x -q -m gpu > gpurun_out/r06i_tests.log 2>&1; echo "tests rc $?"; tail -3 gpurun_out/r06i_tests.log
B="python bench.py --no-cpu-baseline --no-catalogue --no-extra --steps 30 --warmup 5"
run() { tag=$1; shift; env "$@" $B 2> /dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"])' $tag; }
for rep in 1 2; do
run enc_x6 X=1
run enc_fp32 CLSR_ENC_BWD=fp32
done
$B --precision fp32x3 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("fp32x3", d["ms_per_step"])'
bash scripts/prof_step.sh r06i_fp32 | tail -1

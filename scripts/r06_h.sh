#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_rnn_forms_gpu.py tests/test_kernels_gpu.py tests/test_step_gpu.py tests/test_siblings_gpu.py -x -q -m gpu > gpurun_out/r06h_tests.log 2>&1; echo "tests rc $?"; tail -3 gpurun_out/r06h_tests.log
B="python bench.py --no-cpu-baseline --no-catalogue --no-extra --steps 30 --warmup 5"
run() { tag=$1; shift; env "$@" $B 2> /dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"])' $tag; }
for rep in 1 2; do
run x6_fused X=1
run x6_unfused CLSR_NO_RNN_FUSED_PROJ=1
run rnn_fp32 CLSR_RNN_PRODUCTS=fp32
done
$B --precision fp32x3 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("fp32x3", d["ms_per_step"])'
bash scripts/prof_step.sh r06h_fp32 | tail -1

#!/bin/bash
# Round 6, second call: precision="fp32" is the strict mode now -- tests of both fp32-storage modes + step times
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_abort_gpu.py tests/test_att_bwd_x3_gpu.py tests/test_x3_gpu.py tests/test_step_gpu.py tests/test_p2p_gpu.py tests/test_fuzz_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu > gpurun_out/r06b_tests.log 2>&1; echo "tests rc $?"; tail -5 gpurun_out/r06b_tests.log
B="python bench.py --no-cpu-baseline --no-catalogue --no-extra --steps 30 --warmup 5"
for p in fp32 fp32x3 fp32 fp32x3; do $B --precision $p 2> gpurun_out/r06b_$p.err | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"])' $p; done
bash scripts/prof_step.sh r06b_fp32 | tail -1

#!/bin/bash
# attention backward with three pieces (precision="fp32"): kernel tests, step tests, step times, timeline
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_att_bwd_x3_gpu.py tests/test_step_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu > gpurun_out/r06f_tests.log 2>&1; echo "tests rc $?"; tail -3 gpurun_out/r06f_tests.log
B="python bench.py --no-cpu-baseline --no-catalogue --no-extra --steps 30 --warmup 5"
for p in fp32 fp32x3 fp32; do $B --precision $p 2> gpurun_out/r06f_$p.err | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d.get("roofline_att_bwd", {}).get("us_per_launch"))' $p; done
CLSR_ATT_BWD=fp32 $B 2> /dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("att_bwd=fp32", d["ms_per_step"])'
bash scripts/prof_step.sh r06f_fp32 | tail -1

mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests/test_att_bwd_x3_gpu.py -x -q > gpurun_out/r5/x3tests.log 2>&1; echo "x3 tests rc $?"; tail -15 gpurun_out/r5/x3tests.log
timeout 300 python scripts/bench_att_bwd.py 2>&1 | tee gpurun_out/r5/bench_att_bwd.txt
timeout 900 python -m pytest tests/test_step_gpu.py -x -q > gpurun_out/r5/steptests.log 2>&1; echo "step tests rc $?"; tail -8 gpurun_out/r5/steptests.log
for m in x3 fp32 x3 fp32; do CLSR_ATT_BWD=$m timeout 300 python bench.py --no-cpu-baseline --no-catalogue --no-extra --steps 40 2>&1 | grep -E "timed" | sed "s/^/$m /"; done
bash scripts/prof_step.sh r05b_fp32

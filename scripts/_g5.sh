mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests/test_att_bwd_x3_gpu.py -x -q > gpurun_out/r5/x3tests.log 2>&1; echo "x3 tests rc $?"; tail -12 gpurun_out/r5/x3tests.log
timeout 300 python scripts/bench_att_bwd.py 2>&1 | grep -v amdgpu.ids | grep "history\|l0 " | tee gpurun_out/r5/bench_att_bwd4.txt
timeout 900 python -m pytest tests/test_step_gpu.py tests/test_bf16_gpu.py -x -q > gpurun_out/r5/steptests.log 2>&1; echo "step+bf16 tests rc $?"; tail -8 gpurun_out/r5/steptests.log
B="python bench.py --no-cpu-baseline --no-catalogue --no-extra --steps 40"
for rep in 1 2 3; do
echo "default      $($B 2>&1 | grep -E timed)"
echo "nohistbwd    $(CLSR_NO_ATT_HIST_BWD_X3=1 $B 2>&1 | grep -E timed)"
echo "nohistfwd    $(CLSR_NO_ATT_HIST_X3=1 $B 2>&1 | grep -E timed)"
echo "x3gemm all   $(CLSR_X3_GEMM=all $B 2>&1 | grep -E timed)"
done
bash scripts/prof_step.sh r05e_fp32
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r5/gputests2.log 2>&1; echo "all gpu tests rc $?"; tail -8 gpurun_out/r5/gputests2.log

mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests/test_att_bwd_x3_gpu.py tests/test_fuzz_gpu.py -x -q > gpurun_out/r5/x3tests.log 2>&1; echo "x3+fuzz tests rc $?"; tail -5 gpurun_out/r5/x3tests.log
timeout 300 python scripts/bench_att_bwd.py 2>&1 | grep -v amdgpu.ids | grep "l1\|prologue" | tee gpurun_out/r5/bench_att_bwd5.txt
B="python bench.py --no-cpu-baseline --no-catalogue --no-extra --steps 40"
for rep in 1 2 3; do
echo "default      $($B 2>&1 | grep -E timed)"
echo "nol1fwdx6    $(CLSR_NO_ATT_L1_FWD_X6=1 $B 2>&1 | grep -E timed)"
done
bash scripts/prof_step.sh r05f_fp32
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r5/gputests2.log 2>&1; echo "all gpu tests rc $?"; tail -8 gpurun_out/r5/gputests2.log

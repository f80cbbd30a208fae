mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests/test_att_bwd_x3_gpu.py tests/test_step_gpu.py tests/test_fuzz_gpu.py -q > gpurun_out/r5/x3tests.log 2>&1; echo "tests rc $?"; tail -4 gpurun_out/r5/x3tests.log
timeout 300 python scripts/bench_att_bwd.py 2>&1 | grep "att_l0_fwd" 
B="python bench.py --no-cpu-baseline --no-catalogue --no-extra --steps 40"
for rep in 1 2 3; do
echo "default      $($B 2>&1 | grep -E timed)"
echo "nol0fwdx6    $(CLSR_NO_ATT_FWD_X6=1 $B 2>&1 | grep -E timed)"
done
for rep in 1 2; do
echo "dp1 sync-bn  $(CLSR_FORCE_DP=1 $B 2>&1 | grep -E timed)"
echo "dp1 local-bn $(CLSR_FORCE_DP=1 $B --local-bn 2>&1 | grep -E timed)"
echo "single       $($B 2>&1 | grep -E timed)"
done

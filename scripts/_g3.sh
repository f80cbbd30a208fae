mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests/test_att_bwd_x3_gpu.py -x -q > gpurun_out/r5/x3tests.log 2>&1; echo "x3 tests rc $?"; tail -12 gpurun_out/r5/x3tests.log
timeout 300 python scripts/bench_att_bwd.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5/bench_att_bwd2.txt
for v in NODW NOLDS NOSTORE RING2 NOALL; do CLSR_LIB=$PWD/build/abl/lib_x3_$v.so ONLY_L0X3=$v timeout 120 python scripts/bench_att_bwd.py 2>&1 | grep "x3 " | tee -a gpurun_out/r5/bench_att_bwd2.txt; done
timeout 900 python -m pytest tests/test_step_gpu.py -x -q > gpurun_out/r5/steptests.log 2>&1; echo "step tests rc $?"; tail -8 gpurun_out/r5/steptests.log
B="python bench.py --no-cpu-baseline --no-catalogue --no-extra --steps 40"
for rep in 1 2; do
echo "default      $($B 2>&1 | grep -E timed)"
echo "noattfwdx3   $(CLSR_NO_ATT_FWD_X3=1 $B 2>&1 | grep -E timed)"
echo "nox3enc      $(CLSR_NO_X3_ENC=1 $B 2>&1 | grep -E timed)"
echo "x3gemm none  $(CLSR_X3_GEMM= $B 2>&1 | grep -E timed)"
echo "x3gemm all   $(CLSR_X3_GEMM=all $B 2>&1 | grep -E timed)"
echo "exact        $(CLSR_EXACT_PRODUCTS=1 $B 2>&1 | grep -E timed)"
done
bash scripts/prof_step.sh r05c_fp32

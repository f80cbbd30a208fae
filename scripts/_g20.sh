mkdir -p gpurun_out/r5
timeout 1200 python -m pytest tests/test_att_bwd_x3_gpu.py tests/test_step_gpu.py tests/test_fuzz_gpu.py tests/test_fullsize_gpu.py -q > gpurun_out/r5/x3tests.log 2>&1; echo "tests rc $?"; tail -3 gpurun_out/r5/x3tests.log
B="python bench.py --no-cpu-baseline --no-catalogue --no-extra --steps 40"
for rep in 1 2 3; do
echo "default      $($B 2>&1 | grep -E timed)"
echo "noencback    $(CLSR_NO_ENC_BACK_X3=1 $B 2>&1 | grep -E timed)"
done
bash scripts/prof_step.sh r05m_fp32
grep "enc_back\|enc_bwd_fused\|t4_time_inputs_bwd" gpurun_out/r05m_fp32_timeline.txt

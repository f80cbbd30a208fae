mkdir -p gpurun_out/r5
timeout 1200 python -m pytest tests/test_att_bwd_x3_gpu.py tests/test_step_gpu.py tests/test_bf16_gpu.py tests/test_fuzz_gpu.py tests/test_rnn_forms_gpu.py -q > gpurun_out/r5/x3tests.log 2>&1; echo "tests rc $?"; tail -5 gpurun_out/r5/x3tests.log
B="python bench.py --no-cpu-baseline --no-catalogue --no-extra --steps 40"
for rep in 1 2 3; do
echo "default      $($B 2>&1 | grep -E timed)"
echo "noprojx3     $(CLSR_NO_PROJ_X3=1 $B 2>&1 | grep -E timed)"
done
echo "bf16         $($B --precision bf16 2>&1 | grep -E timed)"
echo "kuaishou     $($B --config kuaishou 2>&1 | grep -E timed)"
bash scripts/prof_step.sh r05i_fp32

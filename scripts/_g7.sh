mkdir -p gpurun_out/r5
timeout 1200 python -m pytest tests/test_att_bwd_x3_gpu.py tests/test_fuzz_gpu.py tests/test_bf16_gpu.py tests/test_step_gpu.py tests/test_det_grads_gpu.py -q > gpurun_out/r5/x3tests.log 2>&1; echo "targeted tests rc $?"; tail -6 gpurun_out/r5/x3tests.log
B="python bench.py --no-cpu-baseline --no-catalogue --no-extra --steps 40"
for rep in 1 2 3; do
echo "default      $($B 2>&1 | grep -E timed)"
echo "aux->dw0     $(CLSR_AUX_ALIAS=@dw0 $B 2>&1 | grep -E timed)"
echo "nosplitg2    $(CLSR_NO_SPLIT_G2=1 $B 2>&1 | grep -E timed)"
echo "both         $(CLSR_AUX_ALIAS=@dw0 CLSR_NO_SPLIT_G2=1 $B 2>&1 | grep -E timed)"
done
bash scripts/prof_step.sh r05g_fp32
CLSR_AUX_ALIAS=@dw0 bash scripts/prof_step.sh r05g_fp32_auxdw0
bash scripts/prof_embed.sh r05 > gpurun_out/r5/prof_embed.log 2>&1; tail -12 gpurun_out/r5/prof_embed.log

mkdir -p gpurun_out/r5
timeout 1200 python -m pytest tests/test_det_grads_gpu.py tests/test_kernels_gpu.py tests/test_step_gpu.py -q > gpurun_out/r5/x3tests.log 2>&1; echo "targeted tests rc $?"; tail -4 gpurun_out/r5/x3tests.log
B="python bench.py --no-cpu-baseline --no-catalogue --no-extra --steps 40"
for rep in 1 2 3; do
echo "default      $($B 2>&1 | grep -E timed)"
echo "g2->dw0      $(CLSR_G2_STREAM=@dw0 $B 2>&1 | grep -E timed)"
echo "g2,aux->dw0  $(CLSR_G2_STREAM=@dw0 CLSR_AUX_ALIAS=@dw0 $B 2>&1 | grep -E timed)"
done
CLSR_G2_STREAM=@dw0 bash scripts/prof_step.sh r05h_fp32_g2dw0
bash scripts/prof_embed.sh r05 > gpurun_out/r5/prof_embed.log 2>&1; grep "ss_\|rs_" gpurun_out/r05_embed_kernel_trace.md; grep "segmented" gpurun_out/r05_embed_isolated.txt

mkdir -p gpurun_out/r5
timeout 1500 python -m pytest tests/test_att_chain_bf16_gpu.py tests/test_bf16_gpu.py tests/test_att_bwd_x3_gpu.py -x -q > gpurun_out/r5/h2_tests.log 2>&1; echo "tests rc $?"; tail -15 gpurun_out/r5/h2_tests.log
B="python bench.py --no-cpu-baseline --no-catalogue --no-extra --steps 40 --precision bf16"
for rep in 1 2; do
echo "bf16 chain x1  $($B 2>&1 | grep -E timed)"
echo "bf16 chain old $(CLSR_BF16_CHAIN=old $B 2>&1 | grep -E timed)"
done
bash scripts/prof_step.sh r05b_bf16 --precision bf16
bash scripts/prof_step.sh r05b_fp32

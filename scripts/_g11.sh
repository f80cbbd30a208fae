mkdir -p gpurun_out/r5
timeout 1200 python -m pytest tests/test_bf16_gpu.py tests/test_bf16_tables_gpu.py -x -q > gpurun_out/r5/bf16tests.log 2>&1; echo "bf16 tests rc $?"; tail -6 gpurun_out/r5/bf16tests.log
B="python bench.py --no-cpu-baseline --no-catalogue --no-extra --steps 40 --precision bf16"
for rep in 1 2 3; do
echo "bf16 split    $($B 2>&1 | grep -E timed)"
echo "bf16 nosplit  $(CLSR_BF16_NO_SPLIT_QUERY=1 $B 2>&1 | grep -E timed)"
done
bash scripts/prof_step.sh r05_bf16 --precision bf16

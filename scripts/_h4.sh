mkdir -p gpurun_out/r5
timeout 600 python -m pytest tests/test_att_chain_bf16_gpu.py -x -q 2>&1 | tail -3
B="python bench.py --config catalogue100m --no-cpu-baseline --no-catalogue --no-extra --steps 20 --warmup 3"
for rep in 1 2; do
echo "catalogue default        $($B 2>&1 | grep -E timed)"
echo "catalogue x3 xw          $(CLSR_X3_GEMM=xw^T,xw $B 2>&1 | grep -E 'timed|rror' | head -2)"
done

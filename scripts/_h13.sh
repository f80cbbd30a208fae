timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_step_gpu.py tests/test_bf16_gpu.py -x -q 2>&1 | tail -3
B="python bench.py --no-cpu-baseline --no-catalogue --no-extra --steps 40"
for rep in 1 2 3; do
echo "fp32   $($B 2>&1 | grep -E 'timed|rror' | head -2)"
done
echo "bf16   $($B --precision bf16 2>&1 | grep -E 'timed|rror' | head -2)"
echo "kuaishou $($B --config kuaishou 2>&1 | grep -E 'timed|rror' | head -2)"
bash scripts/prof_step.sh r05f_fp32
grep "att_out_fwd\|att_score_bwd\|att_dy1" gpurun_out/r05f_fp32_timeline.txt

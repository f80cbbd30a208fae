timeout 900 python -m pytest tests/test_att_bwd_x3_gpu.py tests/test_step_gpu.py tests/test_fuzz_gpu.py tests/test_bf16_gpu.py -x -q 2>&1 | tail -3
B="python bench.py --no-cpu-baseline --no-catalogue --no-extra --steps 40"
for rep in 1 2 3; do
echo "tt fused     $($B 2>&1 | grep -E 'timed|rror' | head -2)"
echo "tt side      $(CLSR_NO_PROJ_TT=1 $B 2>&1 | grep -E timed)"
done
echo "kuaishou tt fused $($B --config kuaishou 2>&1 | grep -E 'timed|rror' | head -2)"
echo "kuaishou tt side  $(CLSR_NO_PROJ_TT=1 $B --config kuaishou 2>&1 | grep -E timed)"
bash scripts/prof_step.sh r05d_fp32

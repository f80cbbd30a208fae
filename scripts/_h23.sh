timeout 1200 python -m pytest tests/test_att_bwd_x3_gpu.py tests/test_fullsize_gpu.py tests/test_step_gpu.py -x -q 2>&1 | tail -3
B="python bench.py --config catalogue100m --no-cpu-baseline --no-catalogue --no-extra --steps 20 --warmup 3"
for rep in 1 2 3; do
echo "catalogue n-loop      $($B 2>&1 | grep -E 'timed|rror' | head -2)"
echo "catalogue grid.y      $(CLSR_PROJ_NO_NLOOP=1 $B 2>&1 | grep -E timed)"
done
bash scripts/prof_step.sh r05j_cat --config catalogue100m --steps 8 | tail -1
grep "proj_x3" gpurun_out/r05j_cat_timeline.txt | head -5

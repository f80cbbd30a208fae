timeout 1200 python -m pytest tests/test_att_bwd_x3_gpu.py tests/test_fullsize_gpu.py -x -q 2>&1 | tail -3
B="python bench.py --config catalogue100m --no-cpu-baseline --no-catalogue --no-extra --steps 20 --warmup 3"
for rep in 1 2 3; do
echo "catalogue bwd 2 pieces  $($B 2>&1 | grep -E 'timed|rror' | head -2)"
echo "catalogue bwd 3 pieces  $(CLSR_PROJ_BWD_PIECES=3 $B 2>&1 | grep -E timed)"
done

timeout 1200 python -m pytest tests/test_att_bwd_x3_gpu.py tests/test_kernels_gpu.py tests/test_fullsize_gpu.py tests/test_step_gpu.py tests/test_det_grads_gpu.py -x -q 2>&1 | tail -3
B="python bench.py --config catalogue100m --no-cpu-baseline --no-catalogue --no-extra --steps 20 --warmup 3"
for rep in 1 2; do
echo "catalogue new (proj 512-thread WGs, l0 fwd wave K=128)  $($B 2>&1 | grep -E 'timed|rror' | head -2)"
echo "catalogue proj 256-thread WGs                           $(CLSR_PROJ_WG256=1 $B 2>&1 | grep -E timed)"
echo "catalogue l0 fwd position-tiled                         $(CLSR_NO_L0_FWD_WAVE=1 $B 2>&1 | grep -E timed)"
done
bash scripts/prof_step.sh r05h_cat --config catalogue100m --steps 8 | tail -1

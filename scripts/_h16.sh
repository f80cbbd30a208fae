timeout 900 python -m pytest tests/test_det_grads_gpu.py tests/test_fullsize_gpu.py tests/test_step_gpu.py -x -q 2>&1 | tail -2
for v in default b16 c16 c64b16 c64; do
  if [ $v = default ]; then L=""; else L="CLSR_LIB=$PWD/build/abl/lib_ss$v.so"; fi
  for rep in 1 2; do
    echo "$v item: $(env $L EMBED_SITES=item python scripts/prof_kernels.py embed 2>&1 | grep 'segmented' | sed 's/.*d(hist):/ /' | tr '\n' '|')"
  done
  echo "$v both: $(env $L python scripts/prof_kernels.py embed 2>&1 | grep 'segmented' | sed 's/.*d(hist):/ /' | tr '\n' '|')"
done

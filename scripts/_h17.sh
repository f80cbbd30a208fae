for v in default c16 c64; do
  if [ $v = default ]; then L=""; else L="CLSR_LIB=$PWD/build/abl/lib_ss$v.so"; fi
  for rep in 1 2; do
    echo "$v item: $(env $L EMBED_SITES=item python scripts/prof_kernels.py embed 2>&1 | grep 'segmented' | sed 's/.*d(hist):/ /' | tr '\n' '|')"
  done
  echo "$v both: $(env $L python scripts/prof_kernels.py embed 2>&1 | grep 'segmented' | sed 's/.*d(hist):/ /' | tr '\n' '|')"
done
B="python bench.py --config catalogue100m --no-cpu-baseline --no-catalogue --no-extra --steps 20 --warmup 3"
echo "catalogue default $($B 2>&1 | grep timed)"
echo "catalogue c16     $(CLSR_LIB=$PWD/build/abl/lib_ssc16.so $B 2>&1 | grep timed)"
B1="python bench.py --no-cpu-baseline --no-catalogue --no-extra --steps 40"
echo "taobao default $($B1 2>&1 | grep timed)"
echo "taobao c16     $(CLSR_LIB=$PWD/build/abl/lib_ssc16.so $B1 2>&1 | grep timed)"
echo "taobao default $($B1 2>&1 | grep timed)"
echo "taobao c16     $(CLSR_LIB=$PWD/build/abl/lib_ssc16.so $B1 2>&1 | grep timed)"

B="python bench.py --no-cpu-baseline --no-catalogue --no-extra --steps 40"
for rep in 1 2 3; do
echo "NU 9 early keys   $($B 2>&1 | grep -E 'timed|rror' | head -2)"
echo "NU 4 (before)     $(CLSR_LIB=$PWD/build/abl/lib_attout4.so $B 2>&1 | grep -E 'timed|rror' | head -2)"
echo "NU 9 late keys    $(CLSR_LIB=$PWD/build/abl/lib_attout9late.so $B 2>&1 | grep -E 'timed|rror' | head -2)"
done
echo "bf16 NU 9 early   $($B --precision bf16 2>&1 | grep -E 'timed|rror' | head -2)"
echo "bf16 NU 4         $(CLSR_LIB=$PWD/build/abl/lib_attout4.so $B --precision bf16 2>&1 | grep -E 'timed|rror' | head -2)"

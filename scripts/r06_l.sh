#!/bin/bash
for v in "" ch16 ch64; do
  if [ -n "$v" ]; then export CLSR_LIB=$PWD/build/abl/lib_$v.so; else unset CLSR_LIB; fi
  python bench.py --no-cpu-baseline --no-extra --steps 5 --warmup 2 2>&1 | grep "roofline gather_bwd" | sed "s/^/[$v] /"
done

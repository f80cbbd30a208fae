# A/B of the backward recurrence launch: prefetch of the saved activations vs three waves per SIMD (build/abl/lib_x_*.so)
mkdir -p gpurun_out/r5
for v in base NOPF WPE3; do
  lib=$PWD/build/abl/lib_x_$v.so; [ $v = base ] && lib=$PWD/clsr_amd/libclsr_hip.so
  echo "== $v"; CLSR_LIB=$lib RNN_TILED=1 RNN_FUSED=1 python scripts/bench_rnn.py 2>&1 | grep -E "bwd"
  CLSR_LIB=$lib python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | cut -c1-180
done

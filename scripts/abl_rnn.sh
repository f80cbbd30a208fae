# ablation timings of the split-bf16 Time4LSTM forward (build/abl/lib_x_*.so from scripts/build_variant.sh)
mkdir -p gpurun_out/r5
for v in base NOBAR NOLOAD NOACT NOSTORE NOMFMA NOLDS; do
  lib=$PWD/build/abl/lib_x_$v.so; [ $v = base ] && lib=$PWD/clsr_amd/libclsr_hip.so
  echo "== $v"; CLSR_LIB=$lib CLSR_RNN_PRODUCTS=x3 python scripts/bench_rnn.py 2>&1 | grep -E "fwd  t4|fwd  gru \+"
done

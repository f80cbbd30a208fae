#!/bin/bash
# Timing ablations of the one-wave-per-encoder recurrence kernels: builds variants of libclsr_hip.so whose rnn1.o is
# compiled with one R1_ABL_* macro each (build/abl/), for `CLSR_LIB=build/abl/lib_<v>.so python scripts/bench_rnn.py`.
# Run HERE (cross-compile), the variants travel with the gpurun snapshot.
set -e
cd "$(dirname "$0")/.."
mkdir -p build/abl
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -Iinclude"
others=$(ls build/obj/*.o | grep -v "/rnn1.o")
for v in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS -DR1_ABL_$v -c clsr_amd/csrc/rnn1.hip -o build/abl/rnn1_$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/abl/lib_$v.so $others build/abl/rnn1_$v.o
  echo built build/abl/lib_$v.so
done

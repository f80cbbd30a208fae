#!/bin/bash
# Variant of libclsr_hip.so with extra compiler flags on some sources (diagnosis / ablation builds; the product build is
# python -m clsr_amd.build):  bash scripts/build_variant.sh <name> "<flags>" file1.hip [file2.hip ...]
#   -> build/abl/lib_<name>.so   (use with CLSR_LIB=$PWD/build/abl/lib_<name>.so)
set -e
cd "$(dirname "$0")/.."
name=$1; flags=$2; shift 2
mkdir -p build/abl
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -Iinclude"
objs=$(ls build/obj/*.o)
for f in "$@"; do
  b=$(basename $f .hip)
  /opt/rocm/bin/hipcc $FL $flags -c clsr_amd/csrc/$b.hip -o build/abl/${b}_$name.o
  objs=$(echo "$objs" | grep -v "/$b.o")
  objs="$objs
"build/abl/${b}_$name.o; objs=$(printf "%b" "$objs")
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/abl/lib_$name.so $objs
echo built build/abl/lib_$name.so

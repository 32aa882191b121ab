#!/bin/bash
# tools/build_lib_variant.sh NAME "-DFLAG=.. ..."  ->  build/NAME/libdeclip_hip.so (gemm_v4.hip recompiled with the flags, the other objects reused)
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffast-math -fno-finite-math-only -munsafe-fp-atomics -w"
OBJS=$(ls declip_amd/csrc/*.o | grep -v gemm_v4.o)
mkdir -p build/$NAME
/opt/rocm/bin/hipcc $FLAGS "$@" -c declip_amd/csrc/gemm_v4.hip -o build/$NAME/gemm_v4.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/$NAME/libdeclip_hip.so $OBJS build/$NAME/gemm_v4.o
rm -f build/$NAME/gemm_v4.o

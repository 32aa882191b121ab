#!/bin/bash
# tools/build_lib_variant.sh NAME "-DFLAG=.. ..."  ->  build/NAME/libdeclip_hip.so (gemm_v4.hip recompiled with the flags, the other objects reused)
# DH_VARIANT_SRC=attention (environment): that source instead of gemm_v4
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
SRC=${DH_VARIANT_SRC:-gemm_v4}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffast-math -fno-finite-math-only -munsafe-fp-atomics -w"
OBJS=$(ls declip_amd/csrc/*.o | grep -v /$SRC.o)
mkdir -p build/$NAME
/opt/rocm/bin/hipcc $FLAGS "$@" -c declip_amd/csrc/$SRC.hip -o build/$NAME/$SRC.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/$NAME/libdeclip_hip.so $OBJS build/$NAME/$SRC.o
rm -f build/$NAME/$SRC.o

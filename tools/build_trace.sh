#!/bin/bash
# Trace build of libdeclip_hip.so (gemm_v4.hip with -DV4_TRACE=1): build/trace/libdeclip_hip.so
set -e
cd "$(dirname "$0")/.."
python -m declip_amd.build > /dev/null
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffast-math -fno-finite-math-only -munsafe-fp-atomics -Wno-unused-result -Wno-unused-value"
OBJS=$(ls declip_amd/csrc/*.o | grep -v gemm_v4.o)
mkdir -p build/trace
/opt/rocm/bin/hipcc $FLAGS -DV4_TRACE=1 -c declip_amd/csrc/gemm_v4.hip -o build/trace/gemm_v4.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/trace/libdeclip_hip.so $OBJS build/trace/gemm_v4.o
ls -la build/trace/libdeclip_hip.so

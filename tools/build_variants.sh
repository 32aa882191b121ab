#!/bin/bash
# Build the production library plus the trace build of gemm_v4 (build/trace/libdeclip_hip.so: -DV4_TRACE=1) and the trace tool.
set -e
cd "$(dirname "$0")/.."
python -m declip_amd.build > /dev/null 2>&1
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffast-math -fno-finite-math-only -munsafe-fp-atomics -w"
OBJS=$(ls declip_amd/csrc/*.o | grep -v gemm_v4.o)
mkdir -p build/trace
/opt/rocm/bin/hipcc $FLAGS -DV4_TRACE=${V4_TRACE:-1} -c declip_amd/csrc/gemm_v4.hip -o build/trace/gemm_v4.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/trace/libdeclip_hip.so $OBJS build/trace/gemm_v4.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -w -Iinclude tools/gemm_trace.cpp -Ldeclip_amd -ldeclip_hip -ldl -o tools/gemm_trace
ls -la declip_amd/libdeclip_hip.so build/trace/libdeclip_hip.so tools/gemm_trace

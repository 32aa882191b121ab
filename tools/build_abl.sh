#!/bin/bash
# Build ablation variants of libdeclip_hip.so (only gemm_v4.hip differs): build/abl_<mask>/libdeclip_hip.so
# Use:  LD_LIBRARY_PATH=build/abl_8 tools/gemm_probe 30 quick
set -e
cd "$(dirname "$0")/.."
python -m declip_amd.build > /dev/null
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffast-math -fno-finite-math-only -munsafe-fp-atomics -Wno-unused-result -Wno-unused-value"
OBJS=$(ls declip_amd/csrc/*.o | grep -v gemm_v4.o)
for m in "$@"; do
  mkdir -p build/abl_$m
  ( /opt/rocm/bin/hipcc $FLAGS -DV4_ABL=$m -c declip_amd/csrc/gemm_v4.hip -o build/abl_$m/gemm_v4.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/abl_$m/libdeclip_hip.so $OBJS build/abl_$m/gemm_v4.o ) &
done
wait
ls -la build/abl_*/libdeclip_hip.so

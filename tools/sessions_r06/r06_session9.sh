#!/bin/bash
# round 6, session 9 (V4_COLMAP: each wave owns whole 128-byte lines): V4_WAVE_EPI (wave-private staging of the bf16 epilogue: no workgroup barrier inside the epilogue): GEMM tests + the
# full-width golden step, in-kernel traces, per-shape times against hipBLASLt, same-box step A/B against the same sources with -DV4_WAVE_EPI=0
# and against the library of the round-5 head.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06s9; mkdir -p $O
python -m declip_amd.build > /dev/null 2>&1
bash tools/build_lib_variant.sh wave0 -DV4_WAVE_EPI=0 > /dev/null 2>&1
timeout 1500 python -m pytest tests/test_gpu_gemm_v4.py tests/test_gpu_golden_fullwidth.py tests/test_gpu_graph.py tests/test_gpu_block.py -x -q -p no:cacheprovider > $O/tests.txt 2>&1; tail -4 $O/tests.txt | cut -c1-300
bash tools/build_trace.sh > /dev/null 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -w -Iinclude tools/gemm_trace.cpp -Ldeclip_amd -ldeclip_hip -ldl -o tools/gemm_trace
for sh in "25600 768 768 0" "25600 3072 768 1" "25600 3072 768 2" "25600 768 3072 3" "25600 768 768 3" "22016 512 512 3" "25600 2304 768 0" "25600 768 768 4"; do
  echo "=== $sh"; LD_LIBRARY_PATH=build/trace timeout 120 tools/gemm_trace $sh 0 10
done > $O/trace_epilogue.txt 2>&1
grep -A3 "===" $O/trace_epilogue.txt | cut -c1-260
timeout 600 python tools/bench_hipblaslt.py > $O/gemm_vs_hipblaslt.txt 2>&1; tail -22 $O/gemm_vs_hipblaslt.txt
P="DECLIP_HIP_LIB=$GRAFT_REPO_ROOT/build/prev/libdeclip_hip.so DH_LIB_ALLOW_MISSING=1"
W="DECLIP_HIP_LIB=$GRAFT_REPO_ROOT/build/wave0/libdeclip_hip.so"
bash tools/ab_bench.sh $O/ab "prev:$P" "wave0:$W" "new:" "prev:$P" "wave0:$W" "new:" "wave0:$W" "new:" 2>&1 | tee $O/ab.txt

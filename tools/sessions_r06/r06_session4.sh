#!/bin/bash
# round 6, session 4: V4_PRE_RING (first half of the residual / dGELU epilogue operand fetched by LDS-DMA from inside the last two K-tiles):
# GEMM tests + the full-width golden step, in-kernel traces of the two flavours, per-shape times against hipBLASLt, same-box step A/B.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06s4; mkdir -p $O
python -m declip_amd.build > /dev/null 2>&1
timeout 1500 python -m pytest tests/test_gpu_gemm_v4.py tests/test_gpu_golden_fullwidth.py tests/test_gpu_graph.py tests/test_gpu_kernels.py -x -q -p no:cacheprovider > $O/tests.txt 2>&1; tail -4 $O/tests.txt | cut -c1-300
bash tools/build_trace.sh > /dev/null 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -w -Iinclude tools/gemm_trace.cpp -Ldeclip_amd -ldeclip_hip -ldl -o tools/gemm_trace
for sh in "25600 3072 768 2" "25600 768 3072 3" "25600 768 768 3" "22016 512 512 3" "25600 768 768 0"; do
  echo "=== $sh"; LD_LIBRARY_PATH=build/trace timeout 120 tools/gemm_trace $sh 0 10
done > $O/trace_epilogue.txt 2>&1
grep -A3 "===" $O/trace_epilogue.txt | cut -c1-260
timeout 600 python tools/bench_hipblaslt.py > $O/gemm_vs_hipblaslt.txt 2>&1; tail -22 $O/gemm_vs_hipblaslt.txt
P="DECLIP_HIP_LIB=$GRAFT_REPO_ROOT/build/prev/libdeclip_hip.so DH_LIB_ALLOW_MISSING=1"
bash tools/ab_bench.sh $O/ab "prev:$P" "new:" "prev:$P" "new:" "prev:$P" "new:" 2>&1 | tee $O/ab.txt

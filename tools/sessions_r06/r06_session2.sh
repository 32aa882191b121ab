#!/bin/bash
# round 6, session 2: (a) is the epilogue's pre-operand stall (residual / dGELU aux loads) burst contention or latency?  The same
# flavours with 30 / 90 / 300 tiles in flight; (b) dynamic tile distribution in the one-GPU two-stream step (round 5 gave it the fast hand-over).
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06s2; mkdir -p $O
python -m declip_amd.build > /dev/null 2>&1
bash tools/build_trace.sh > /dev/null 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -w -Iinclude tools/gemm_trace.cpp -Ldeclip_amd -ldeclip_hip -ldl -o tools/gemm_trace
for sh in "2560 768 768 3" "7680 768 768 3" "21760 768 768 3" "25600 768 768 3" "65536 768 768 3" "2560 3072 768 2" "7680 3072 768 2" "25600 3072 768 2" "2560 768 768 0" "2560 3072 768 1"; do
  echo "=== $sh"; LD_LIBRARY_PATH=build/trace timeout 120 tools/gemm_trace $sh 0 10
done > $O/trace_burst.txt 2>&1
grep -A3 "===" $O/trace_burst.txt | cut -c1-250
bash tools/ab_bench.sh $O/ab "static:" "dyn:DH_V4_DYNAMIC=1" "static:" "dyn:DH_V4_DYNAMIC=1" 2>&1 | tee $O/ab.txt

#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r3g
mkdir -p $O
for shape in "25600 768 768 0 0" "25600 768 768 0 1" "2304 768 25600 0 1" "25600 3072 768 1 0"; do
  echo "=== $shape (V4_SCHED=1, steady-state stamps in K-tile 2)" >> $O/trace.txt
  LD_LIBRARY_PATH=$PWD/build/trace timeout 120 tools/gemm_trace $shape 14 2>&1 | head -6 >> $O/trace.txt
done
cut -c1-330 $O/trace.txt

#!/bin/bash
# round 3, GPU session C: in-kernel s_memtime traces of gemm_v4, round-2 kernel (7 stamps per tile) vs the continuous K-tile stream
# (10 stamps per tile), same box
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r3c
mkdir -p $O
for shape in "25600 768 768 0" "22016 2048 512 0" "25600 3072 768 1" "25600 768 3072 0"; do
  echo "=== $shape (round-2 kernel)" >> $O/trace.txt
  LD_LIBRARY_PATH=$PWD/build/trace_old timeout 120 tools/gemm_trace $shape 0 7 2>&1 | head -12 >> $O/trace.txt
  echo "=== $shape (K-tile stream)" >> $O/trace.txt
  LD_LIBRARY_PATH=$PWD/build/trace timeout 120 tools/gemm_trace $shape 0 10 2>&1 | head -12 >> $O/trace.txt
done
cat $O/trace.txt

#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r3e
mkdir -p $O
for shape in "25600 768 768 0" "22016 2048 512 0"; do
  echo "=== $shape (fine stamps)" >> $O/trace.txt
  LD_LIBRARY_PATH=$PWD/build/trace timeout 120 tools/gemm_trace $shape 0 18 2>&1 | head -8 >> $O/trace.txt
done
cut -c1-420 $O/trace.txt

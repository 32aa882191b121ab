#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r3h
mkdir -p $O
for v in t_q t_q_nodma t_q_nofrag t_q_nomfma t_q_mfmaonly t_s0 t_s0_nodma; do
  for shape in "25600 768 768 0 0" "25600 768 768 0 1"; do
    echo "=== $v $shape" >> $O/trace.txt
    LD_LIBRARY_PATH=$PWD/build/$v timeout 120 tools/gemm_trace $shape 10 2>&1 | head -4 | tail -2 >> $O/trace.txt
  done
done
cut -c1-260 $O/trace.txt

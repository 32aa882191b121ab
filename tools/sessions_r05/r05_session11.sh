#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05s11; mkdir -p $O
bash tools/ab_bench.sh $O/ab "v4:" "v6auto:DH_GEMM_V6=2" "v4:" "v6auto:DH_GEMM_V6=2" "v6all:DH_GEMM_V6=1" "v4:" "v6auto:DH_GEMM_V6=2" 2>&1 | tee $O/ab.txt

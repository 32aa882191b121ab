#!/bin/bash
# round 6, session 28: the M = 512 GEMMs of the pooled last blocks, per kernel family, in a captured chain (tools/bench_small_gemm.py)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06s28; mkdir -p $O
python -m declip_amd.build > /dev/null 2>&1
python tools/bench_small_gemm.py > $O/small_gemm.txt 2>&1; cat $O/small_gemm.txt | grep -v amdgpu

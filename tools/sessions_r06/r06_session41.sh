#!/bin/bash
# round 6, session 41: the projection GEMMs of the pooled features per kernel family (tools/bench_proj_gemm.py)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06s41; mkdir -p $O
python -m declip_amd.build > /dev/null 2>&1
python tools/bench_proj_gemm.py 2>&1 | grep -v amdgpu > $O/proj.txt; cat $O/proj.txt

#!/bin/bash
# round 6, session 46: tile order of the persistent GEMM (DH_V4_GROUP_M: row groups sharing their B tiles in L2; default 1 for forward / dX, 8 for dW) re-measured in the step
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06s46; mkdir -p $O
python -m declip_amd.build > /dev/null 2>&1
bash tools/ab_bench.sh $O/ab "default:" "gm2:DH_V4_GROUP_M=2" "gm4:DH_V4_GROUP_M=4" "gm8:DH_V4_GROUP_M=8" "default:" "gm2:DH_V4_GROUP_M=2" "gm4:DH_V4_GROUP_M=4" "gm8:DH_V4_GROUP_M=8" > $O/ab.txt 2>&1; cat $O/ab.txt

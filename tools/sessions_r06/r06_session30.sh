#!/bin/bash
# round 6, session 30: K-slices of the grouped weight-gradient launch (DH_V4_GROUP_SPLIT; default: cost model) measured in the step
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06s30; mkdir -p $O
python -m declip_amd.build > /dev/null 2>&1
bash tools/ab_bench.sh $O/ab "auto:" "s1:DH_V4_GROUP_SPLIT=1" "s2:DH_V4_GROUP_SPLIT=2" "s3:DH_V4_GROUP_SPLIT=3" "s4:DH_V4_GROUP_SPLIT=4" "auto:" "s1:DH_V4_GROUP_SPLIT=1" "s2:DH_V4_GROUP_SPLIT=2" "s3:DH_V4_GROUP_SPLIT=3" "s4:DH_V4_GROUP_SPLIT=4" > $O/ab.txt 2>&1; cat $O/ab.txt

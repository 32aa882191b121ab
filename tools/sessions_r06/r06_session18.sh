#!/bin/bash
# round 6, session 18: priority of the image tower's stream (DH_SIDE_PRIORITY=-1: its workgroups first when both streams have work pending), captured and eager step
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06s18; mkdir -p $O
python -m declip_amd.build > /dev/null 2>&1
bash tools/ab_bench.sh $O/ab "equal:" "imgfirst:DH_SIDE_PRIORITY=-1" "equal:" "imgfirst:DH_SIDE_PRIORITY=-1" "equal:" "imgfirst:DH_SIDE_PRIORITY=-1" > $O/ab.txt 2>&1; cat $O/ab.txt
bash tools/ab_bench.sh $O/ab_eager "equal:" "imgfirst:DH_SIDE_PRIORITY=-1" "equal:" "imgfirst:DH_SIDE_PRIORITY=-1" -- --graph 0 > $O/ab_eager.txt 2>&1; cat $O/ab_eager.txt

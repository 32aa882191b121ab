#!/bin/bash
# round 6, session 34: the reduce pass of the image tower's grouped weight-gradient launches on another stream (DH_DW_REDUCE_STREAM=1: the text tower's, 2: its own): A/B in the step + gradient equality
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06s34; mkdir -p $O
python -m declip_amd.build > /dev/null 2>&1
bash tools/ab_bench.sh $O/ab "inline:" "text:DH_DW_REDUCE_STREAM=1" "own:DH_DW_REDUCE_STREAM=2" "inline:" "text:DH_DW_REDUCE_STREAM=1" "own:DH_DW_REDUCE_STREAM=2" "inline:" "text:DH_DW_REDUCE_STREAM=1" "own:DH_DW_REDUCE_STREAM=2" > $O/ab.txt 2>&1; cat $O/ab.txt
bash tools/ab_bench.sh $O/ab_eager "inline:" "text:DH_DW_REDUCE_STREAM=1" "own:DH_DW_REDUCE_STREAM=2" -- --graph 0 > $O/ab_eager.txt 2>&1; cat $O/ab_eager.txt
tail -2 $O/ab/own_2.err

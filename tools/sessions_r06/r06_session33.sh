#!/bin/bash
# round 6, session 33: balanced rounds for ONE tower's GEMMs only (2: text tower shapes, 3: image tower shapes), in the step
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06s33; mkdir -p $O
python -m declip_amd.build > /dev/null 2>&1
bash tools/ab_bench.sh $O/ab "full:" "text:DH_V4_BALANCED=2" "image:DH_V4_BALANCED=3" "full:" "text:DH_V4_BALANCED=2" "image:DH_V4_BALANCED=3" "full:" "text:DH_V4_BALANCED=2" "image:DH_V4_BALANCED=3" > $O/ab.txt 2>&1; cat $O/ab.txt

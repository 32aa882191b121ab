#!/bin/bash
# round 6, session 35: LayerNorm forward block cap with the row-pair kernel, in the step
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06s35; mkdir -p $O
python -m declip_amd.build > /dev/null 2>&1
bash tools/ab_bench.sh $O/ab "cap1024:" "cap2048:DH_LN_FWD_CAP=2048" "cap512:DH_LN_FWD_CAP=512" "cap1024:" "cap2048:DH_LN_FWD_CAP=2048" "cap512:DH_LN_FWD_CAP=512" "cap1024:" "cap2048:DH_LN_FWD_CAP=2048" "cap512:DH_LN_FWD_CAP=512" > $O/ab.txt 2>&1; cat $O/ab.txt

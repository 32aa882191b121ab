#!/bin/bash
# round 6, session 37: prefetch depth of the packed text attention kernels once more, in the step, three interleaved rounds
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06s37; mkdir -p $O
python -m declip_amd.build > /dev/null 2>&1
bash tools/ab_bench.sh $O/ab "pf1:" "pf2:DH_ATTN_PF=2" "pf3:DH_ATTN_PF=3" "bwd2:DH_ATTN_PF_BWD=2" "pf1:" "pf2:DH_ATTN_PF=2" "pf3:DH_ATTN_PF=3" "bwd2:DH_ATTN_PF_BWD=2" "pf1:" "pf2:DH_ATTN_PF=2" "pf3:DH_ATTN_PF=3" "bwd2:DH_ATTN_PF_BWD=2" > $O/ab.txt 2>&1; cat $O/ab.txt

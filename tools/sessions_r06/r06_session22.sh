#!/bin/bash
# round 6, session 22: grid-size knobs of the HBM-bound kernels measured IN THE STEP (they were tuned alone on the chip): LayerNorm forward block cap, attention workgroups per CU
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06s22; mkdir -p $O
python -m declip_amd.build > /dev/null 2>&1
bash tools/ab_bench.sh $O/ab "base:" "lncap512:DH_LN_FWD_CAP=512" "lncap2048:DH_LN_FWD_CAP=2048" "attncap2:DH_ATTN_WG_CAP=2" "attncap3:DH_ATTN_WG_CAP=3" "attncap8:DH_ATTN_WG_CAP=8" "base:" "lncap512:DH_LN_FWD_CAP=512" "lncap2048:DH_LN_FWD_CAP=2048" "attncap2:DH_ATTN_WG_CAP=2" "attncap3:DH_ATTN_WG_CAP=3" "attncap8:DH_ATTN_WG_CAP=8" > $O/ab.txt 2>&1; cat $O/ab.txt

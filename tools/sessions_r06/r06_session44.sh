#!/bin/bash
# round 6, session 44: two more choices that were tuned alone on the chip, re-measured in the step: few-tile K-split of the M = 512 GEMMs (DH_V4_TAIL), batched LayerNorm-gradient reduce (DH_LN_BATCH)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06s44; mkdir -p $O
python -m declip_amd.build > /dev/null 2>&1
bash tools/ab_bench.sh $O/ab "base:" "notail:DH_V4_TAIL=0" "nolnbatch:DH_LN_BATCH=0" "base:" "notail:DH_V4_TAIL=0" "nolnbatch:DH_LN_BATCH=0" "base:" "notail:DH_V4_TAIL=0" "nolnbatch:DH_LN_BATCH=0" > $O/ab.txt 2>&1; cat $O/ab.txt

#!/bin/bash
# round 6, session 45: attention length buckets on / off for the multi-view families (DeCLIP, SLIP, FILIP), same box
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06s45; mkdir -p $O
python -m declip_amd.build > /dev/null 2>&1
for m in declip slip filip; do
  bash tools/ab_bench.sh $O/ab_$m "one:" "buckets:DH_ATTN_BUCKETS=1" "one:" "buckets:DH_ATTN_BUCKETS=1" -- --model $m --steps 12 > $O/ab_$m.txt 2>&1; echo "--- $m"; cat $O/ab_$m.txt
done

#!/bin/bash
# round 6, session 42: the two length buckets of the packed text attention against one launch per call (DH_ATTN_BUCKETS=0), in the step
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06s42; mkdir -p $O
python -m declip_amd.build > /dev/null 2>&1
bash tools/ab_bench.sh $O/ab "buckets:" "one:DH_ATTN_BUCKETS=0" "buckets:" "one:DH_ATTN_BUCKETS=0" "buckets:" "one:DH_ATTN_BUCKETS=0" "buckets:" "one:DH_ATTN_BUCKETS=0" "buckets:" "one:DH_ATTN_BUCKETS=0" "buckets:" "one:DH_ATTN_BUCKETS=0" > $O/ab.txt 2>&1; cat $O/ab.txt

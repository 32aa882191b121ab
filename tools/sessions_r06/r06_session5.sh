#!/bin/bash
# round 6, session 5: (a) packed text attention with two pairs prefetched ahead / more workgroups per CU (alone + in the step), LayerNorm
# forward with 4 rows per wave; (b) the multi-GPU rank's default path on one GPU after round 6 (library communicator by default, no sleep
# before the capture): 6 runs; (c) kernel trace of the default step: which small launches are left.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06s5; mkdir -p $O
python -m declip_amd.build > /dev/null 2>&1
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -p no:cacheprovider -k "attn or attention or layernorm or ln_" > $O/tests.txt 2>&1; tail -3 $O/tests.txt | cut -c1-200
for v in "pf1:DH_ATTN_PF=1" "pf2:DH_ATTN_PF=2" "pf3:DH_ATTN_PF=3" "pf2cap6:DH_ATTN_PF=2 DH_ATTN_WG_CAP=6" "pf1cap6:DH_ATTN_PF=1 DH_ATTN_WG_CAP=6" "bwdpf2:DH_ATTN_PF_BWD=2" "bwdpf3:DH_ATTN_PF_BWD=3" "lnr4:DH_LN_FWD_R=4"; do
  name=${v%%:*}; envs=${v#*:}
  echo "--- $name ($envs)"; env $envs BENCH_SMALL=attn_txt,ln timeout 300 python tools/bench_small.py 2>&1 | grep -v amdgpu.ids
done > $O/small_variants.txt 2>&1; cat $O/small_variants.txt
bash tools/ab_bench.sh $O/ab "pf1:DH_ATTN_PF=1" "pf2:DH_ATTN_PF=2" "pf2cap6:DH_ATTN_PF=2 DH_ATTN_WG_CAP=6" "pf1:DH_ATTN_PF=1" "pf2:DH_ATTN_PF=2" "pf2cap6:DH_ATTN_PF=2 DH_ATTN_WG_CAP=6" 2>&1 | tee $O/ab.txt
for i in 1 2 3 4 5 6; do
  DH_DIST_FORCE=1 timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-loss-delta --no-roofline > $O/force_$i.json 2> $O/force_$i.err
  python - $O/force_$i.json <<'PY' || tail -3 $O/force_$i.err
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("forced one-rank group: %9.1f pairs/s %8.3f ms/step graph %s fallback %s comm_native %s" % (d["value"], d["ms_per_step"], d["config"].get("step_graph"), d.get("graph_fallback"), d["config"].get("comm_native")))
PY
done 2>&1 | tee $O/force.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/trace -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-loss-delta --no-roofline > $GRAFT_REPO_ROOT/$O/trace.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find $O/trace -name "*.db" | head -1); python tools/rocpd_stats.py $DB > $O/stats.txt 2>&1; head -70 $O/stats.txt | cut -c1-150
rm -rf $O/trace

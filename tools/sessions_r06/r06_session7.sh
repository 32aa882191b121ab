#!/bin/bash
# round 6, session 7: the optimizer step overlapped with the backward pass (optim.FlatAdamW.enable_overlap): tests (bit-exact against the
# one-launch step; eager and captured), same-box A/B in the CLIP step, and the same for a multi-GPU rank's configuration (per-bucket step
# behind the bucket's all-reduce on the communication stream).
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06s7; mkdir -p $O
python -m declip_amd.build > /dev/null 2>&1
timeout 1500 python -m pytest tests/test_gpu_opt_overlap.py tests/test_gpu_clip.py tests/test_gpu_solver.py tests/test_gpu_graph.py -x -q -p no:cacheprovider > $O/tests.txt 2>&1; tail -5 $O/tests.txt | cut -c1-300
bash tools/ab_bench.sh $O/ab "end:DH_OPT_OVERLAP=0" "overlap:DH_OPT_OVERLAP=1" "end:DH_OPT_OVERLAP=0" "overlap:DH_OPT_OVERLAP=1" "end:DH_OPT_OVERLAP=0" "overlap:DH_OPT_OVERLAP=1" 2>&1 | tee $O/ab.txt
bash tools/ab_bench.sh $O/abf "f_end:DH_DIST_FORCE=1 DH_OPT_OVERLAP=0" "f_overlap:DH_DIST_FORCE=1 DH_OPT_OVERLAP=1" "f_end:DH_DIST_FORCE=1 DH_OPT_OVERLAP=0" "f_overlap:DH_DIST_FORCE=1 DH_OPT_OVERLAP=1" 2>&1 | tee $O/ab_forced.txt
for m in declip slip filip defilip; do
  for v in 0 1; do
    DH_OPT_OVERLAP=$v timeout 600 python bench.py --model $m --steps 10 --warmup 3 --no-cpu-baseline --no-loss-delta --no-roofline > $O/${m}_$v.json 2> $O/${m}_$v.err
    python - $O/${m}_$v.json $m $v <<'PY' || tail -3 $O/${m}_$v.err
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-8s overlap %s: %9.1f pairs/s %8.3f ms/step graph %s" % (sys.argv[2], sys.argv[3], d["value"], d["ms_per_step"], d["config"].get("step_graph")))
PY
  done
done 2>&1 | tee $O/families.txt

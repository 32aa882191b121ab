#!/bin/bash
# round 6, session 6: per-element bf16 gates of the non-GEMM kernels at the step's shapes (margins recorded), the one-rank RCCL tests after
# the switch to the library communicator, the solver's captured step as a multi-GPU rank, and 20 captured one-rank runs without the sleep.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06s6; mkdir -p $O
python -m declip_amd.build > /dev/null 2>&1
DH_MARGIN_RECORD=$GRAFT_REPO_ROOT/$O/margins_new.json timeout 1200 python -m pytest tests/test_gpu_bf16_elementwise.py -x -q -p no:cacheprovider -s > $O/tests_elementwise.txt 2>&1; grep -E "^margin|passed|failed|Error|assert" $O/tests_elementwise.txt | cut -c1-220 | tail -40
timeout 1500 python -m pytest tests/test_gpu_dist.py tests/test_gpu_bench_fallback.py tests/test_gpu_solver.py -x -q -p no:cacheprovider > $O/tests_dist.txt 2>&1; tail -5 $O/tests_dist.txt | cut -c1-300
for v in plain forced plain forced; do
  if [ $v == forced ]; then DH_DIST_FORCE=1 timeout 600 python tools/solver_step_bench.py 2>&1 | grep "solver step"; else timeout 600 python tools/solver_step_bench.py 2>&1 | grep "solver step"; fi
done | tee $O/solver_step.txt
ok=0
for i in $(seq 1 20); do
  DH_DIST_FORCE=1 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-loss-delta --no-roofline > $O/run20.json 2> $O/run20.err
  rc=$?
  python - $O/run20.json $rc <<'PY' && ok=$((ok+1))
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    good = int(sys.argv[2]) == 0 and d["config"]["step_graph"] == 1 and d["graph_fallback"] is None and d["config"]["comm_native"] == 1
    print("run: rc %s %9.1f pairs/s graph %s fallback %s comm_native %s" % (sys.argv[2], d["value"], d["config"]["step_graph"], d["graph_fallback"], d["config"]["comm_native"]))
    sys.exit(0 if good else 1)
except Exception as e:
    print("run: rc %s FAILED %r" % (sys.argv[2], e)); sys.exit(1)
PY
done > $O/captured_one_rank_20.txt 2>&1
echo "captured one-rank runs without the pre-capture sleep: $ok of 20 clean" | tee -a $O/captured_one_rank_20.txt

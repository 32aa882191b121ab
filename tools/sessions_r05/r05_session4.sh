#!/bin/bash
# round 5, GPU session 4: multi-GPU rank's default path on one GPU (captured step over a one-rank RCCL group), fallbacks, SLIP under the graph
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05s4; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_bench_fallback.py tests/test_gpu_graph.py tests/test_gpu_dist.py tests/test_gpu_solver.py -q -p no:cacheprovider -x > $O/tests.txt 2>&1; tail -15 $O/tests.txt
bash tools/ab_bench.sh $O/ab "plain:" "force:DH_DIST_FORCE=1" "plain:" "force:DH_DIST_FORCE=1" "force_eager:DH_DIST_FORCE=1 DH_STEP_GRAPH=0" 2>&1 | tee $O/ab.txt
DH_DIST_FORCE=1 python bench.py --no-cpu-baseline --no-loss-delta --no-roofline > $O/bench_force.json 2> $O/bench_force.err; cat $O/bench_force.json
python bench.py --model slip --no-cpu-baseline --no-roofline > $O/bench_slip.json 2> $O/bench_slip.err; cat $O/bench_slip.json | cut -c1-300
python bench.py --model slip --graph 0 --no-cpu-baseline --no-roofline > $O/bench_slip_eager.json 2> $O/bench_slip_eager.err; cat $O/bench_slip_eager.json | cut -c1-300

#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05s5; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_bench_fallback.py tests/test_gpu_graph.py tests/test_gpu_dist.py tests/test_gpu_solver.py -q -p no:cacheprovider > $O/tests.txt 2>&1; tail -15 $O/tests.txt | cut -c1-250
bash tools/ab_bench.sh $O/ab "plain:" "dyn:DH_V4_DYNAMIC=1" "force:DH_DIST_FORCE=1" "force_1bucket:DH_DIST_FORCE=1 DH_BUCKET_MB=2048" "plain:" "force:DH_DIST_FORCE=1" "force_eager:DH_DIST_FORCE=1 DH_STEP_GRAPH=0" "force_native:DH_DIST_FORCE=1 DH_COMM_NATIVE=1" 2>&1 | tee $O/ab.txt

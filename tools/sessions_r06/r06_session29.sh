#!/bin/bash
# round 6, session 29: GraphedStep fallback branches (capture failure local / on a peer, first-replay failure on a peer / local), bench fallback, one-rank RCCL step
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06s29; mkdir -p $O
python -m declip_amd.build > /dev/null 2>&1
python -m pytest tests/test_gpu_graph.py tests/test_gpu_bench_fallback.py tests/test_gpu_dist.py tests/test_bench_launch.py -q -x > $O/tests_full.txt 2>&1; grep -n "Fatal\|Current thread" -A 6 $O/tests_full.txt | head -30; tail -4 $O/tests_full.txt | cut -c1-200

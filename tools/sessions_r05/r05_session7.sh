#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05s7; mkdir -p $O
for i in 1 2 3 4 5 6 7 8; do
DH_DIST_FORCE=1 python bench.py --steps 3 --warmup 1 --batch 256 --no-cpu-baseline --no-loss-delta --no-roofline > $O/out_$i.json 2> $O/err_$i.txt; echo "rc=$? $(grep -c terminate $O/err_$i.txt)"
done
timeout 900 python -m pytest tests/test_gpu_bench_fallback.py tests/test_gpu_dist.py -q -p no:cacheprovider 2>&1 | tail -4 | cut -c1-300

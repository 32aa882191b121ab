#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05s10; mkdir -p $O
for i in 1 2 3 4 5 6 7 8 9 10; do
DH_DIST_FORCE=1 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-loss-delta --no-roofline > $O/out_$i.json 2> $O/err_$i.txt; echo "rc=$? terminate=$(grep -c terminate $O/err_$i.txt) $(cut -c1-90 $O/out_$i.json)"
done

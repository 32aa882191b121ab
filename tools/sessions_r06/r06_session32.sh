#!/bin/bash
# round 6, session 32: balanced rounds (DH_V4_BALANCED=1: a tile list longer than the chip gets ceil(items / rounds) workgroups instead of 256): GEMM tests, alone, in the step
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06s32; mkdir -p $O
python -m declip_amd.build > /dev/null 2>&1
DH_V4_BALANCED=1 python -m pytest tests/test_gpu_gemm_v4.py -q -x 2>&1 | tail -1 > $O/tests.txt; cat $O/tests.txt
for v in 0 1; do echo "--- DH_V4_BALANCED=$v"; DH_V4_BALANCED=$v python tools/bench_hipblaslt.py --out $O/hb_$v.txt > /dev/null 2>&1; tail -1 $O/hb_$v.txt; done
bash tools/ab_bench.sh $O/ab "full:" "balanced:DH_V4_BALANCED=1" "full:" "balanced:DH_V4_BALANCED=1" "full:" "balanced:DH_V4_BALANCED=1" > $O/ab.txt 2>&1; cat $O/ab.txt

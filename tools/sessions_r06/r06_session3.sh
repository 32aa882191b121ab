#!/bin/bash
# round 6, session 3: first check of the round's changes on the GPU -- housekeeping (tail modes / v6 out of the library, switches read
# once), DPP wave reductions in LayerNorm, the two attention length buckets forked onto two streams; kernel tests, the HBM-bound
# kernels alone, and a same-box A/B of the step against the library of the round-5 head (build/prev).
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06s3; mkdir -p $O
python -m declip_amd.build > /dev/null 2>&1
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_gemm_v4.py tests/test_gpu_graph.py tests/test_gpu_block.py -x -q -p no:cacheprovider > $O/tests.txt 2>&1; tail -5 $O/tests.txt | cut -c1-300
BENCH_SMALL=attn_txt,ln timeout 300 python tools/bench_small.py > $O/small_new.txt 2>&1; cat $O/small_new.txt
DH_ATTN_FORK=0 BENCH_SMALL=attn_txt timeout 300 python tools/bench_small.py > $O/small_nofork.txt 2>&1; cat $O/small_nofork.txt
P="DECLIP_HIP_LIB=$GRAFT_REPO_ROOT/build/prev/libdeclip_hip.so DH_LIB_ALLOW_MISSING=1"
bash tools/ab_bench.sh $O/ab "prev:$P" "new:" "nofork:DH_ATTN_FORK=0" "prev:$P" "new:" "nofork:DH_ATTN_FORK=0" 2>&1 | tee $O/ab.txt

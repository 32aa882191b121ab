#!/bin/bash
# round 5, GPU session 2: in-kernel tail fix-up v2 (sc1 buffer traffic, fix-up deferred to the end of the kernel) + fast hand-over under the dynamic distribution
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05s2; mkdir -p $O
export DH_MARGIN_RECORD=$GRAFT_REPO_ROOT/$O/bf16_margins.json
timeout 900 python -m pytest tests/test_gpu_gemm_v4.py -q -s -p no:cacheprovider > $O/gemm_v4_tests.txt 2>&1; tail -12 $O/gemm_v4_tests.txt
timeout 900 python -m pytest tests/test_gpu_golden_fullwidth.py -q -s -p no:cacheprovider -k bf16 > $O/fullwidth_bf16.txt 2>&1; tail -5 $O/fullwidth_bf16.txt
unset DH_MARGIN_RECORD
for v in "3:DH_V4_TAIL=3" "1s2:DH_V4_TAIL_SMAX=2" "1s3:DH_V4_TAIL_SMAX=3" "1s5:DH_V4_TAIL_SMAX=5" "1s3k12:DH_V4_TAIL_SMAX=3 DH_V4_TAIL_MINK=12" "1s2k12:DH_V4_TAIL_SMAX=2 DH_V4_TAIL_MINK=12"; do
  n=${v%%:*}; e=${v#*:}
  env $e python tools/bench_hipblaslt.py --only-n 768 --out $O/yard_tail$n.txt > /dev/null 2>&1; echo "== $v"; cat $O/yard_tail$n.txt | grep -v "^#" 
done
bash tools/ab_bench.sh $O/ab "tail3:DH_V4_TAIL=3" "s2:DH_V4_TAIL_SMAX=2" "s3:DH_V4_TAIL_SMAX=3" "s5:DH_V4_TAIL_SMAX=5" "tail3:DH_V4_TAIL=3" "s3:DH_V4_TAIL_SMAX=3" "dyn:DH_V4_DYNAMIC=1 DH_V4_TAIL=3" "dyn:DH_V4_DYNAMIC=1 DH_V4_TAIL=3" 2>&1 | tee $O/ab.txt

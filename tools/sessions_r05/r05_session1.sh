#!/bin/bash
# round 5, GPU session 1: in-kernel tail fix-up (tests + A/B) and the bf16 margins of the full-width fixtures
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05s1; mkdir -p $O
export DH_MARGIN_RECORD=$GRAFT_REPO_ROOT/$O/bf16_margins.json
timeout 900 python -m pytest tests/test_gpu_gemm_v4.py -q -s -p no:cacheprovider > $O/gemm_v4_tests.txt 2>&1; tail -5 $O/gemm_v4_tests.txt
timeout 900 python -m pytest tests/test_gpu_golden_fullwidth.py -q -s -p no:cacheprovider -k bf16 > $O/fullwidth_bf16.txt 2>&1; tail -5 $O/fullwidth_bf16.txt
unset DH_MARGIN_RECORD
bash tools/ab_bench.sh $O/ab "tail3:DH_V4_TAIL=3" "tail1:" "tail2:DH_V4_TAIL=2" "tail1k12:DH_V4_TAIL_MINK=12 DH_V4_TAIL_SMAX=3" "tail3:DH_V4_TAIL=3" "tail1:" "tail1s4:DH_V4_TAIL_SMAX=4" 2>&1 | tee $O/ab.txt
python tools/bench_hipblaslt.py --only-n 768 --out $O/yard_tail1.txt > /dev/null 2>&1; cat $O/yard_tail1.txt
DH_V4_TAIL=3 python tools/bench_hipblaslt.py --only-n 768 --out $O/yard_tail3.txt > /dev/null 2>&1; cat $O/yard_tail3.txt
DH_V4_TAIL_MINK=12 DH_V4_TAIL_SMAX=3 python tools/bench_hipblaslt.py --only-n 768 --out $O/yard_tail1_k12s3.txt > /dev/null 2>&1; cat $O/yard_tail1_k12s3.txt
DH_V4_TAIL_MINK=12 DH_V4_TAIL_SMAX=2 python tools/bench_hipblaslt.py --only-n 768 --out $O/yard_tail1_k12s2.txt > /dev/null 2>&1; cat $O/yard_tail1_k12s2.txt

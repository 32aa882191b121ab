#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05s8; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_gemm_v4.py tests/test_gpu_block.py tests/test_gpu_clip.py tests/test_gpu_golden_fullwidth.py tests/test_gpu_graph.py -q -p no:cacheprovider > $O/tests.txt 2>&1; tail -8 $O/tests.txt | cut -c1-300
bash tools/ab_bench.sh $O/ab "ft0:DH_FIRST_TOUCH=0" "ft1:" "ft0:DH_FIRST_TOUCH=0" "ft1:" "ft0:DH_FIRST_TOUCH=0" "ft1:" 2>&1 | tee $O/ab.txt
python tools/prof_copies.py > $O/prof_copies.txt 2>&1; tail -80 $O/prof_copies.txt

#!/bin/bash
# round 5, GPU session 3: QuickGELU' as the aux output of the forward epilogue (DH_EPI_DGELU = a plain multiply): tests + A/B against the previous library
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05s3; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_gemm_v4.py tests/test_gpu_kernels.py tests/test_gpu_block.py tests/test_gpu_golden_fullwidth.py tests/test_gpu_clip.py -q -p no:cacheprovider > $O/tests.txt 2>&1; tail -8 $O/tests.txt
bash tools/ab_bench.sh $O/ab "prev:DECLIP_HIP_LIB=$GRAFT_REPO_ROOT/build/prev/libdeclip_hip.so" "new:" "prev:DECLIP_HIP_LIB=$GRAFT_REPO_ROOT/build/prev/libdeclip_hip.so" "new:" "prev:DECLIP_HIP_LIB=$GRAFT_REPO_ROOT/build/prev/libdeclip_hip.so" "new:" 2>&1 | tee $O/ab.txt
python tools/bench_hipblaslt.py --out $O/yard_new.txt > /dev/null 2>&1; cat $O/yard_new.txt

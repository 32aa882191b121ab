#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05s13; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_clip.py tests/test_gpu_block.py -q -p no:cacheprovider -k "pooled or clip" > $O/tests.txt 2>&1; tail -3 $O/tests.txt | cut -c1-200
bash tools/ab_bench.sh $O/ab "prev:DECLIP_HIP_LIB=$GRAFT_REPO_ROOT/build/prev/libdeclip_hip.so" "new:" "prev:DECLIP_HIP_LIB=$GRAFT_REPO_ROOT/build/prev/libdeclip_hip.so" "new:" "prev:DECLIP_HIP_LIB=$GRAFT_REPO_ROOT/build/prev/libdeclip_hip.so" "new:" 2>&1 | tee $O/ab.txt

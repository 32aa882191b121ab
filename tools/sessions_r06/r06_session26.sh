#!/bin/bash
# round 6, session 26: final attention variant (forward staged for all, backward staged for dense sequences only): tests + A/B against the piece-store build
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06s26; mkdir -p $O
python -m declip_amd.build > /dev/null 2>&1
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bf16_elementwise.py tests/test_gpu_block.py tests/test_gpu_resnet_intake_packed.py tests/test_gpu_clip.py tests/test_gpu_graph.py -q -k "attn or attention or block or packed or clip or graph" 2>&1 | tail -1 > $O/tests.txt; cat $O/tests.txt
V="DECLIP_HIP_LIB=build/attn_piece/libdeclip_hip.so"
bash tools/ab_bench.sh $O/ab "line:" "piece:$V" "line:" "piece:$V" "line:" "piece:$V" > $O/ab.txt 2>&1; cat $O/ab.txt

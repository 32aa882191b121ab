#!/bin/bash
# round 6, session 25: attention forward AND backward outputs as whole 128-byte rows through LDS (ATTN_LINE_OUT=1) against 8-byte pieces (build/attn_piece): tests, alone, in the step
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06s25; mkdir -p $O
python -m declip_amd.build > /dev/null 2>&1
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bf16_elementwise.py tests/test_gpu_block.py tests/test_gpu_resnet_intake_packed.py tests/test_gpu_clip.py -q -k "attn or attention or block or packed or clip" 2>&1 | tail -1 > $O/tests.txt; cat $O/tests.txt
V="DECLIP_HIP_LIB=build/attn_piece/libdeclip_hip.so"
for v in "" "$V" "" "$V"; do echo "--- ${v:-line}"; env $v BENCH_SMALL=attn_img,attn_txt python tools/bench_small.py 2>&1 | grep "attn" | head -6; done > $O/attn_alone.txt; cat $O/attn_alone.txt
bash tools/ab_bench.sh $O/ab "line:" "piece:$V" "line:" "piece:$V" "line:" "piece:$V" > $O/ab.txt 2>&1; cat $O/ab.txt

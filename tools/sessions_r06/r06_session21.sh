#!/bin/bash
# round 6, session 21: LayerNorm backward, d = 768, one and a half chunks per lane (16 + 8 bytes: no masked lanes): tests, kernel alone, A/B in the step
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06s21; mkdir -p $O
python -m declip_amd.build > /dev/null 2>&1
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bf16_elementwise.py tests/test_gpu_block.py -q -k "layernorm or ln or block" 2>&1 | tail -1 > $O/tests.txt; cat $O/tests.txt
for v in 1 0 1 0; do echo "--- DH_LN_BWD_HALF=$v"; DH_LN_BWD_HALF=$v BENCH_SMALL=ln python tools/bench_small.py 2>&1 | grep "LN"; done > $O/ln_alone.txt; cat $O/ln_alone.txt
bash tools/ab_bench.sh $O/ab "half:" "masked:DH_LN_BWD_HALF=0" "half:" "masked:DH_LN_BWD_HALF=0" "half:" "masked:DH_LN_BWD_HALF=0" > $O/ab.txt 2>&1; cat $O/ab.txt

#!/bin/bash
# round 6, session 20: LayerNorm BACKWARD on row pairs (d = 768): tests, kernel alone, A/B in the step (1: 2 waves per SIMD, no spills; 3: 3 waves per SIMD, 55 spilled registers; 0: per-row kernel)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06s20; mkdir -p $O
python -m declip_amd.build > /dev/null 2>&1
for v in 1 3; do DH_LN_BWD_PAIR=$v python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bf16_elementwise.py tests/test_gpu_block.py -q -k "layernorm or ln or block" 2>&1 | tail -1; done > $O/tests.txt; cat $O/tests.txt
for v in 1 3 0; do echo "--- DH_LN_BWD_PAIR=$v"; DH_LN_BWD_PAIR=$v BENCH_SMALL=ln python tools/bench_small.py 2>&1 | grep "LN"; done > $O/ln_alone.txt; cat $O/ln_alone.txt
bash tools/ab_bench.sh $O/ab "pair2:" "pair3:DH_LN_BWD_PAIR=3" "perrow:DH_LN_BWD_PAIR=0" "pair2:" "pair3:DH_LN_BWD_PAIR=3" "perrow:DH_LN_BWD_PAIR=0" "pair2:" "pair3:DH_LN_BWD_PAIR=3" "perrow:DH_LN_BWD_PAIR=0" > $O/ab.txt 2>&1; cat $O/ab.txt

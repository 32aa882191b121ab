#!/bin/bash
# round 6, session 19: LayerNorm forward on row PAIRS (d = 768: three full 1 KB requests per pair instead of four half-masked ones): tests, kernel alone, A/B in the step
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06s19; mkdir -p $O
python -m declip_amd.build > /dev/null 2>&1
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bf16_elementwise.py tests/test_gpu_block.py -q -k "layernorm or ln or block" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
for v in 1 0; do echo "--- DH_LN_FWD_PAIR=$v"; DH_LN_FWD_PAIR=$v BENCH_SMALL=ln python tools/bench_small.py 2>&1 | grep "LN"; done > $O/ln_alone.txt; cat $O/ln_alone.txt
bash tools/ab_bench.sh $O/ab "pair:" "perrow:DH_LN_FWD_PAIR=0" "pair:" "perrow:DH_LN_FWD_PAIR=0" "pair:" "perrow:DH_LN_FWD_PAIR=0" > $O/ab.txt 2>&1; cat $O/ab.txt

#!/bin/bash
# round 6, session 13: vectorised positional-embedding gradient of the packed text tower: parity tests + A/B against the previous library
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06s13; mkdir -p $O
python -m declip_amd.build > /dev/null 2>&1
python -m pytest tests/test_gpu_kernels.py -q -k "packed or embed" > $O/tests.txt 2>&1; tail -5 $O/tests.txt
python -m pytest tests/test_gpu_bf16_elementwise.py tests/test_gpu_engine.py -q -x > $O/tests2.txt 2>&1; tail -5 $O/tests2.txt
P="DECLIP_HIP_LIB=build/prev/libdeclip_hip.so DH_LIB_ALLOW_MISSING=1"
bash tools/ab_bench.sh $O/ab "new:" "prev:$P" "new:" "prev:$P" "new:" "prev:$P" > $O/ab.txt 2>&1; cat $O/ab.txt

#!/bin/bash
# round 6, session 40: both directions of the InfoNCE backward in ONE launch when b == B (DH_NCE_BWD_MERGED=1, default): tests + A/B in the step
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06s40; mkdir -p $O
python -m declip_amd.build > /dev/null 2>&1
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_clip.py tests/test_gpu_graph.py -q -k "nce or infonce or clip or graph" 2>&1 | tail -1 > $O/tests.txt; cat $O/tests.txt
bash tools/ab_bench.sh $O/ab "merged:" "two:DH_NCE_BWD_MERGED=0" "merged:" "two:DH_NCE_BWD_MERGED=0" "merged:" "two:DH_NCE_BWD_MERGED=0" > $O/ab.txt 2>&1; cat $O/ab.txt

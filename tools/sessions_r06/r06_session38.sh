#!/bin/bash
# round 6, session 38: label cache in the loss: parity tests + dispatch count
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06s38; mkdir -p $O
python -m declip_amd.build > /dev/null 2>&1
python -m pytest tests/test_gpu_clip.py tests/test_gpu_graph.py tests/test_gpu_solver.py tests/test_gpu_dist.py -q -x 2>&1 | tail -2 > $O/tests.txt; cat $O/tests.txt

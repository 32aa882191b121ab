#!/bin/bash
# round 6, session 43: default without attention length buckets: parity tests (both settings) + dispatch count
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06s43; mkdir -p $O
python -m declip_amd.build > /dev/null 2>&1
python -m pytest tests/test_gpu_golden_fullwidth.py tests/test_gpu_graph.py tests/test_gpu_resnet_intake_packed.py tests/test_gpu_bf16_elementwise.py tests/test_gpu_block.py -q -x 2>&1 | tail -2 > $O/tests.txt; cat $O/tests.txt

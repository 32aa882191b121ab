#!/bin/bash
# round 6, session 11: the per-element gates (margins recorded), then the whole -m gpu suite on the current sources
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06s11; mkdir -p $O
python -m declip_amd.build > /dev/null 2>&1
DH_MARGIN_RECORD=$GRAFT_REPO_ROOT/$O/margins_new.json timeout 900 python -m pytest tests/test_gpu_bf16_elementwise.py -q -p no:cacheprovider -s 2>&1 | grep -E "^margin|passed|failed|Error" | cut -c1-200 | tee $O/tests_elementwise.txt
timeout 2400 python -m pytest tests/ -q -m gpu -p no:cacheprovider > $O/gpu_suite.txt 2>&1; tail -15 $O/gpu_suite.txt | cut -c1-300

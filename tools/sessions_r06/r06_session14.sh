#!/bin/bash
# round 6, session 14: the torch kernels left in the CLIP step and the host lines that launch them (launch diet)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06s14; mkdir -p $O
python -m declip_amd.build > /dev/null 2>&1
python tools/torch_ops_in_step.py > $O/torch_ops.txt 2>&1; cut -c1-260 $O/torch_ops.txt | tail -120

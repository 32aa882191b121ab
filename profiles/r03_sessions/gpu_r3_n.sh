#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r3n
timeout 300 python tools/diag_pipeline.py > gpurun_out/r3n/diag.txt 2>&1
cat gpurun_out/r3n/diag.txt | grep -v amdgpu.ids

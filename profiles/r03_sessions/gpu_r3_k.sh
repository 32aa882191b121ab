#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r3k
mkdir -p $O
timeout 300 python tools/prof_copies.py > $O/copies.txt 2>&1
tail -120 $O/copies.txt

#!/bin/bash
# round 6, session 36: "CU-compact" text tower: block cap of the text tower's LayerNorm / attention launches (a persistent GEMM workgroup of the image tower needs a CU with nothing else on it), in the step
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06s36; mkdir -p $O
python -m declip_amd.build > /dev/null 2>&1
bash tools/ab_bench.sh $O/ab "base:" "c64:DH_TEXT_COMPACT=64" "c128:DH_TEXT_COMPACT=128" "c256:DH_TEXT_COMPACT=256" "base:" "c64:DH_TEXT_COMPACT=64" "c128:DH_TEXT_COMPACT=128" "c256:DH_TEXT_COMPACT=256" "base:" "c64:DH_TEXT_COMPACT=64" "c128:DH_TEXT_COMPACT=128" "c256:DH_TEXT_COMPACT=256" > $O/ab.txt 2>&1; cat $O/ab.txt

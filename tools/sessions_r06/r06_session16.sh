#!/bin/bash
# round 6, session 16: the image tower's weight-gradient launches deferred to the text tower's stream (DH_DW_DEFER=1): A/B in the captured and the eager step
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06s16; mkdir -p $O
python -m declip_amd.build > /dev/null 2>&1
bash tools/ab_bench.sh $O/ab "inline:DH_DW_DEFER=0" "defer:DH_DW_DEFER=1" "inline:DH_DW_DEFER=0" "defer:DH_DW_DEFER=1" "inline:DH_DW_DEFER=0" "defer:DH_DW_DEFER=1" > $O/ab.txt 2>&1; cat $O/ab.txt
bash tools/ab_bench.sh $O/ab_eager "inline:DH_DW_DEFER=0" "defer:DH_DW_DEFER=1" -- --graph 0 > $O/ab_eager.txt 2>&1; cat $O/ab_eager.txt
for f in $O/ab/inline_0.json $O/ab/defer_1.json; do python -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d.get('loss'), d.get('peak_mem_gb'), d['config'])"; done
tail -3 $O/ab/defer_1.err

#!/bin/bash
# round 6, session 39: the round's net effect on the headline, same box: final sources against the previous commit (git archive HEAD~1 under build/prev_tree), interleaved
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06s39; mkdir -p $O
python -m declip_amd.build > /dev/null 2>&1
for i in 1 2 3 4; do
  python bench.py --no-cpu-baseline --no-loss-delta --no-roofline --steps 30 > $O/r06_$i.json 2> $O/r06_$i.err
  (cd build/prev_tree && python bench.py --no-cpu-baseline --no-loss-delta --no-roofline --steps 30 > ../../$O/prev_$i.json 2> ../../$O/prev_$i.err)
done
for f in $O/r06_*.json $O/prev_*.json; do python -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], d['value'], d['ms_per_step'])"; done

#!/bin/bash
# round 6, session 15: pooled-attention backward zeroes its own gap rows (no dkv fill launch), temperature exp/clamp moved before the towers:
# parity tests + A/B against the previous commit (library AND host code differ: the "old" arm is `git archive HEAD` + its library under build/prev_tree)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06s15; mkdir -p $O
python -m declip_amd.build > /dev/null 2>&1
python -m pytest tests/test_gpu_resnet_intake_packed.py tests/test_gpu_clip.py tests/test_gpu_graph.py tests/test_gpu_bf16_elementwise.py -q -x > $O/tests.txt 2>&1; tail -5 $O/tests.txt
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --no-loss-delta --no-roofline --steps 30 > $O/new_$i.json 2> $O/new_$i.err
  (cd build/prev_tree && python bench.py --no-cpu-baseline --no-loss-delta --no-roofline --steps 30 > ../../$O/old_$i.json 2> ../../$O/old_$i.err)
done
for f in $O/new_*.json $O/old_*.json; do python -c "
import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'])"; done

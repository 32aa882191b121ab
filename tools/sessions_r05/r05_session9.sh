#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05s9; mkdir -p $O
for v in "r2c1024:" "r4c1024:DH_LN_FWD_R=4" "r2c2048:DH_LN_FWD_CAP=2048" "r4c512:DH_LN_FWD_R=4 DH_LN_FWD_CAP=512" "r2c512:DH_LN_FWD_CAP=512" "r4c2048:DH_LN_FWD_R=4 DH_LN_FWD_CAP=2048" "r2c1536:DH_LN_FWD_CAP=1536" "r2c768:DH_LN_FWD_CAP=768"; do
  n=${v%%:*}; e=${v#*:}
  echo "== $n"; env $e BENCH_SMALL=ln python tools/bench_small.py 2>&1 | grep "LN fwd"
done

#!/bin/bash
# GPU session L: do the two tower streams share the chip better with dynamic tile distribution / a prioritised image tower?
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() { echo "== $1"; shift; for i in 1 2; do env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loss-delta --no-roofline 2>&1 | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('   %.1f pairs/s  %.2f ms/step  loss %.4f' % (j['value'], j['ms_per_step'], j['loss']))"; done; }
{
run "default" DH_X=0
run "DH_V4_DYNAMIC=1" DH_V4_DYNAMIC=1
run "DH_SIDE_PRIORITY=-1" DH_SIDE_PRIORITY=-1
run "DH_V4_DYNAMIC=1 DH_SIDE_PRIORITY=-1" DH_V4_DYNAMIC=1 DH_SIDE_PRIORITY=-1
run "DH_V4_DYNAMIC=2" DH_V4_DYNAMIC=2
run "default again" DH_X=0
} 2>&1 | tee gpurun_out/ab_stream_sharing.txt

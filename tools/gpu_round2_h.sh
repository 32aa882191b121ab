#!/bin/bash
# GPU session H: attention kernels that skip fully masked blocks: tests, kernel table, smoke()
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_resnet_intake_packed.py tests/test_gpu_clip.py -m gpu -q -k "attn or attention or packed or clip_fp32 or bf16" > gpurun_out/pytest_attn.txt 2>&1; echo "rc=$?" >> gpurun_out/pytest_attn.txt
tail -5 gpurun_out/pytest_attn.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.txt 2>&1; tail -3 gpurun_out/smoke.txt
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
DH_TOWER_STREAMS=0 timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_h -o trace -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-loss-delta --graph 0 > $R/gpurun_out/prof_h.log 2>&1
DB=$(find $R/gpurun_out/prof_h -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $DB > $R/gpurun_out/stats_h.txt 2>&1
rm -rf $R/gpurun_out/prof_h
grep -E "attn_|TOTAL" $R/gpurun_out/stats_h.txt | cut -c1-150
cd $R; for i in 1 2; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loss-delta --no-roofline 2>&1 | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('   %.1f pairs/s  %.2f ms/step  loss %.4f' % (j['value'], j['ms_per_step'], j['loss']))"; done

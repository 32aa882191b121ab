#!/bin/bash
# GPU session I: communicator context (1 rank), InfoNCE D > 512 on the matrix pipe, sorted embedding gradient, v5 vs v4 test;
# kernel table + CLIP / DeCLIP lines
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_dist.py -m gpu -q -k "one_rank" > gpurun_out/pytest_comm.txt 2>&1; echo "rc=$?" >> gpurun_out/pytest_comm.txt
tail -15 gpurun_out/pytest_comm.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_gemm_v4.py tests/test_gpu_clip.py tests/test_gpu_golden_fullwidth.py -m gpu -q -k "infonce or embed or v5 or clip_fp32 or bf16 or declip or filip" > gpurun_out/pytest_i.txt 2>&1; echo "rc=$?" >> gpurun_out/pytest_i.txt
tail -8 gpurun_out/pytest_i.txt
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
DH_TOWER_STREAMS=0 timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_i -o trace -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-loss-delta --graph 0 > $R/gpurun_out/prof_i.log 2>&1
DB=$(find $R/gpurun_out/prof_i -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $DB > $R/gpurun_out/stats_i.txt 2>&1
rm -rf $R/gpurun_out/prof_i
grep -E "embed|nce_|TOTAL" $R/gpurun_out/stats_i.txt | cut -c1-150
cd $R
for m in clip declip; do for i in 1 2; do echo "== $m"; timeout 300 python bench.py --model $m --steps 20 --warmup 5 --no-cpu-baseline --no-loss-delta --no-roofline 2>&1 | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('   %.1f pairs/s  %.2f ms/step  loss %.4f' % (j['value'], j['ms_per_step'], j['loss']))"; done; done

#!/bin/bash
# round 6: the evidence set of profiles/r06_* from ONE session (kernel table, PMC passes, per-shape GEMM table, yardstick, small kernels, bench lines of all families)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06f; mkdir -p $O
export DH_HEAD=$(cat .head_for_gpurun 2>/dev/null)
DH_BENCH_GEMM_TABLE=$O/gemm_table_clip.txt python bench.py > $O/bench_clip.json 2> $O/bench_clip.err; cut -c1-400 $O/bench_clip.json
python bench.py --pipeline 1 --no-cpu-baseline --no-loss-delta --no-roofline > $O/bench_clip_pipeline.json 2> $O/bench_clip_pipeline.err
python bench.py --graph 0 --no-cpu-baseline --no-loss-delta --no-roofline > $O/bench_clip_eager.json 2> $O/bench_clip_eager.err
DH_DIST_FORCE=1 python bench.py --no-cpu-baseline --no-loss-delta --no-roofline > $O/bench_clip_force.json 2> $O/bench_clip_force.err
for m in declip slip filip defilip; do python bench.py --model $m --no-cpu-baseline > $O/bench_$m.json 2> $O/bench_$m.err; done
python bench.py --model clip_r50 --dtype fp32 --no-cpu-baseline > $O/bench_r50_fp32.json 2> $O/bench_r50_fp32.err
for f in $O/bench_*.json; do python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d.get('roofline') or {}
    print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], 'frac', r.get('frac'), 'step_exec', r.get('step_mfma_frac_executed'), 'graph', d['config'].get('step_graph'))
except Exception as e: print(sys.argv[1], 'FAILED', e)
PY
done
python tools/bench_hipblaslt.py --out $O/gemm_vs_hipblaslt.txt > /dev/null 2>&1; tail -3 $O/gemm_vs_hipblaslt.txt
BENCH_SMALL=all python tools/bench_small.py > $O/small_kernels.txt 2>&1
bash tools/profile_step.sh > $O/profile_step.log 2>&1
cp gpurun_out/prof/stats.txt $O/clip_kernel_stats.txt; cp gpurun_out/prof/pmc_summary.txt $O/clip_pmc_summary.txt
head -30 $O/clip_kernel_stats.txt | cut -c1-160
grep "^JSON " gpurun_out/prof/pmc_summary.txt | tail -1 | cut -c6- > $O/pmc_traffic_clip_b512.json; cat $O/pmc_traffic_clip_b512.json
# FILIP: kernel table of its step (towers on one stream, eager: per-kernel durations alone on the chip)
cd /tmp && export TMPDIR=/tmp
DH_TOWER_STREAMS=0 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/trace_filip -o trace -- python $GRAFT_REPO_ROOT/bench.py --model filip --steps 4 --warmup 2 --no-cpu-baseline --no-loss-delta --no-roofline --graph 0 > $GRAFT_REPO_ROOT/$O/trace_filip.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find $O/trace_filip -name "*.db" | head -1); python tools/rocpd_stats.py $DB > $O/filip_kernel_stats.txt 2>&1; head -24 $O/filip_kernel_stats.txt | cut -c1-160
rm -rf $O/trace_filip
# the torch kernels left in the step
python tools/torch_ops_in_step.py > $O/torch_ops.txt 2>&1
# dispatches / idle time of the default (two-stream, captured) step
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/trace2 -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-loss-delta --no-roofline > $GRAFT_REPO_ROOT/$O/trace2.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find $O/trace2 -name "*.db" | head -1); python tools/rocpd_stats.py $DB > $O/dispatches.txt 2>&1; tail -25 $O/dispatches.txt | cut -c1-200
rm -rf $O/trace2 gpurun_out/prof/trace

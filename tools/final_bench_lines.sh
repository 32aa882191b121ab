cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/f1
# headline line as the driver runs it + per-shape GEMM table
DH_BENCH_GEMM_TABLE=gpurun_out/f1/gemm_table_clip.txt python bench.py > gpurun_out/f1/bench_clip.json 2> gpurun_out/f1/bench_clip.err
python bench.py --pipeline 1 --no-cpu-baseline --no-loss-delta --no-roofline > gpurun_out/f1/bench_clip_pipeline.json 2> gpurun_out/f1/bench_clip_pipeline.err
python bench.py --graph 0 --no-cpu-baseline --no-loss-delta --no-roofline > gpurun_out/f1/bench_clip_eager.json 2> gpurun_out/f1/bench_clip_eager.err
for m in declip slip filip defilip; do python bench.py --model $m --no-cpu-baseline > gpurun_out/f1/bench_$m.json 2> gpurun_out/f1/bench_$m.err; done
python bench.py --model clip_r50 --dtype fp32 --no-cpu-baseline > gpurun_out/f1/bench_r50_fp32.json 2> gpurun_out/f1/bench_r50_fp32.err
BENCH_SMALL=all python tools/bench_small.py > gpurun_out/f1/small_kernels.txt 2>&1
for f in gpurun_out/f1/bench_*.json; do python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d.get('roofline') or {}
    print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], 'frac', r.get('frac'), 'enq', d.get('host_enqueue_ms_empty_queue'))
except Exception as e: print(sys.argv[1], 'FAILED', e)
PY
done

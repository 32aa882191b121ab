#!/bin/bash
# round 6: copy the outputs of tools/r06_final_profile.sh (gpurun_out/r06f) into profiles/r06_* with a header line each (run in the build container after the gpurun call)
cd "$(dirname "$0")/.."
H=$(cat .head_for_gpurun); F=gpurun_out/r06f; P=profiles
{ echo "# round 6, session F (head $H, tools/r06_final_profile.sh): python bench.py -- the line as the driver runs it (CLIP ViT-B/32, b = 512, bf16, 1 x MI355X), then the same step fed by the input pipeline, eager, and as a multi-GPU rank runs it (DH_DIST_FORCE=1: one-rank RCCL group, library communicator, captured)"
  echo "## python bench.py"; cat $F/bench_clip.json
  echo "## python bench.py --pipeline 1 --no-cpu-baseline --no-loss-delta --no-roofline"; cat $F/bench_clip_pipeline.json
  echo "## python bench.py --graph 0 ..."; cat $F/bench_clip_eager.json
  echo "## DH_DIST_FORCE=1 python bench.py ...   (per_rank_ms / allreduce_exposed_ms / allgather_ms / bucket_mb / comm_native: the attribution fields of a --gpus N line)"; cat $F/bench_clip_force.json; } > $P/r06_bench_clip.txt
for m in declip slip filip defilip; do { echo "# round 6, session F (head $H): python bench.py --model $m --no-cpu-baseline"; cat $F/bench_$m.json; } > $P/r06_bench_$m.txt; done
{ echo "# round 6, session F (head $H): python bench.py --model clip_r50 --dtype fp32 --no-cpu-baseline   (BASELINE.json configs[0], the reference's CPU-runnable case)"; cat $F/bench_r50_fp32.json; } > $P/r06_bench_r50_fp32.txt
{ echo "# round 6, session F (head $H): rocprofv3 --kernel-trace of \`DH_TOWER_STREAMS=0 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-loss-delta --graph 0\` (tools/profile_step.sh; towers on ONE stream, eager: per-kernel durations = the kernel alone on the chip; 12 steps incl. the roofline leg's instrumented ones)"; cat $F/clip_kernel_stats.txt; } > $P/r06_clip_kernel_stats.txt
{ echo "# round 6, session F (head $H): PMC passes of the same step (tools/profile_step.sh: one rocprofv3 --pmc pass per counter group, tools/pmc_summary.py)"; cat $F/clip_pmc_summary.txt; } > $P/r06_clip_pmc_summary.txt
cp $F/pmc_traffic_clip_b512.json $P/r06_pmc_traffic_clip_b512.json
{ echo "# round 6, session F (head $H): per-shape GEMM table of the CLIP step (DH_BENCH_GEMM_TABLE, HIP-event brackets on the launch stream, towers on one stream)"; cat $F/gemm_table_clip.txt; } > $P/r06_gemm_table_clip.txt
{ echo "# round 6, session F (head $H): HBM-bound kernels at their in-step shapes (BENCH_SMALL=all python tools/bench_small.py), alone on the chip"; grep -v "amdgpu.ids" $F/small_kernels.txt; } > $P/r06_small_kernels.txt
{ echo "# round 6, session F (head $H): FILIP step (b = 256, B = 256, towers on one stream, eager), rocprofv3 --kernel-trace: \`DH_TOWER_STREAMS=0 python bench.py --model filip --steps 4 --warmup 2 --graph 0\` (9 steps incl. the roofline leg)"; cat $F/filip_kernel_stats.txt; } > $P/r06_filip_kernel_stats.txt
{ echo "# round 6, session F (head $H): the torch (ATen) kernels left in one eager CLIP step (tools/torch_ops_in_step.py; the two dkv fills of the pooled last blocks -- 78 + 45 MB, 34 + 9 us -- are gone: attn_pooled_bwd zeroes its gap rows)"; grep -v "amdgpu.ids\|Warning\|_warn_once" $F/torch_ops.txt; } > $P/r06_torch_ops_in_step.txt
{ echo "# round 6, session F (head $H): dispatches and idle time of the DEFAULT CLIP step (two tower streams, captured hipGraph), rocprofv3 --kernel-trace of"; echo "# \`python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-loss-delta --no-roofline\` (tools/rocpd_stats.py per-step section)"; grep -A 20 "^per step" $F/dispatches.txt; echo "# -> 477 dispatches per step (round 5: 504; VERDICT r5's target of <= 440 is NOT reached -- see DESIGN.md s7 for what the remaining launches are and why merging them buys no time: the GPU is idle 0.04 ms of a step)"; echo; echo "# kernel table of the same trace (in-step durations: overlapped kernels stretch each other)"; head -60 $F/dispatches.txt; } > $P/r06_dispatches.txt
for f in $F/bench_*.json; do python - $f <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d.get('roofline') or {}
print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], 'frac', r.get('frac'), 'achieved', r.get('achieved'), 'step_exec', r.get('step_mfma_frac_executed'))
PY
done

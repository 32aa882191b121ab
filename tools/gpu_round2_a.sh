#!/bin/bash
# GPU session A of round 2: the whole `-m gpu` suite (no xfail anywhere), the default bench line, the per-shape GEMM table,
# DeCLIP / R50 lines, and the rocprofv3 kernel table of the new default (packed captions + pooled last block).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.txt
tail -5 gpurun_out/pytest_gpu.txt
DH_BENCH_GEMM_TABLE=gpurun_out/gemm_table_clip.txt timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_clip.txt 2>gpurun_out/bench_clip.err
tail -1 gpurun_out/bench_clip.txt | cut -c1-600
timeout 300 python bench.py --model declip --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_declip.txt 2>&1
timeout 300 python bench.py --model declip --steps 10 --warmup 3 --no-cpu-baseline --text-packed 0 --pooled-last 0 > gpurun_out/bench_declip_dense.txt 2>&1
timeout 300 python bench.py --model clip_r50 --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_r50_bf16.txt 2>&1
timeout 600 bash tools/profile_step.sh > gpurun_out/profile_step.log 2>&1

#!/bin/bash
# GPU session T: bench.py as the driver launches it for N = 2 (two ranks sharing the one GPU of the test box: gloo collectives on CUDA tensors)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
for m in clip declip; do
  DH_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --model $m --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_w2_$m.txt 2>&1; echo "rc=$?" >> gpurun_out/bench_w2_$m.txt
  grep '^{' gpurun_out/bench_w2_$m.txt | cut -c1-400; tail -3 gpurun_out/bench_w2_$m.txt | cut -c1-300
done

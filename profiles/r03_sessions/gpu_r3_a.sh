#!/bin/bash
# round 3, GPU session A: the new parity / graph / RCCL-capture tests + this round's baseline bench line
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r3a
O=gpurun_out/r3a
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 900 python -m pytest tests/test_gpu_graph.py tests/test_gpu_solver.py -x -q -s -m gpu -k "multiview or prefetcher_leaves" ) > $O/t_graph.log 2>&1
echo "graph rc=$?" >> $O/t_graph.log
( timeout 900 python -m pytest tests/test_gpu_clip.py tests/test_gpu_golden_fullwidth.py -q -s -m gpu -k "bf16" ) > $O/t_bf16.log 2>&1
echo "bf16 rc=$?" >> $O/t_bf16.log
( timeout 900 python -m pytest tests/test_gpu_dist.py -q -s -m gpu -k "one_rank or filip_two" ) > $O/t_dist.log 2>&1
echo "dist rc=$?" >> $O/t_dist.log
( timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_hipemu_kernels.py -q -m gpu -k "filip or nn_bank or maxsim or nn_search" ) > $O/t_ieee.log 2>&1
echo "ieee rc=$?" >> $O/t_ieee.log
( DH_BENCH_GEMM_TABLE=$O/gemm_table_clip.txt timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline ) > $O/bench_clip.log 2>&1
tail -3 $O/t_graph.log $O/t_bf16.log $O/t_dist.log $O/t_ieee.log
tail -2 $O/bench_clip.log

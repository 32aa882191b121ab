#!/bin/bash
# GPU session J: does the epilogue of the persistent GEMM shrink when few CUs store at once? (in-kernel s_memtime trace);
# + re-check of the tests touched since session I
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export LD_LIBRARY_PATH=$(pwd)/build/trace:$LD_LIBRARY_PATH
for cfg in "2560 768 768 0" "25600 768 768 0" "65536 768 768 0" "2560 2048 512 0" "22016 2048 512 0" "2560 3072 768 1" "25600 3072 768 1"; do
  echo "=== $cfg"; timeout 120 build/trace/gemm_trace $cfg 2>&1 | head -14
done > gpurun_out/trace_epilogue.txt 2>&1
cat gpurun_out/trace_epilogue.txt | cut -c1-260
unset LD_LIBRARY_PATH
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "infonce or embed" > gpurun_out/pytest_j.txt 2>&1; echo "rc=$?" >> gpurun_out/pytest_j.txt
tail -4 gpurun_out/pytest_j.txt
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
DH_TOWER_STREAMS=0 timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_j -o trace -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-loss-delta --graph 0 > $R/gpurun_out/prof_j.log 2>&1
DB=$(find $R/gpurun_out/prof_j -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $DB > $R/gpurun_out/stats_j.txt 2>&1
rm -rf $R/gpurun_out/prof_j
grep -E "embed|TOTAL" $R/gpurun_out/stats_j.txt | cut -c1-150

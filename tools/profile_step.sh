#!/bin/bash
# rocprofv3 profile of the bench step on the GPU box (run through gpurun from the repo root):
#   1. kernel trace -> --stats style table (tools/rocpd_stats.py)          -> gpurun_out/prof/stats.txt
#   2. PMC passes (counters only, one pass per counter group as the guide prescribes) -> gpurun_out/prof/pmc_*.csv
# Usage: tools/profile_step.sh [bench args]
set -x
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="${@:---steps 4 --warmup 2 --no-cpu-baseline --no-loss-delta --graph 0}"
# kernel table with the towers on ONE stream: per-kernel durations = the kernel alone on the chip (a trace of the two-stream
# run is not informative: rocprofv3 serialises the dispatches it times)
DH_TOWER_STREAMS=0 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $ROOT/bench.py $ARGS > $OUT/trace.log 2>&1
DB=$(find $OUT/trace -name "*.db" | head -1)
if [ -n "$DB" ]; then python $ROOT/tools/rocpd_stats.py $DB > $OUT/stats.txt 2>&1; fi
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$tag -o pmc -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-loss-delta --no-roofline --graph 0 > $OUT/pmc_$tag.log 2>&1
done
python $ROOT/tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt 2>&1
tail -40 $OUT/pmc_summary.txt
head -40 $OUT/stats.txt

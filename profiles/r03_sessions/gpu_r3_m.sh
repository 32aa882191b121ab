#!/bin/bash
# round 3, GPU session M: FILIP graph test (noise floor), the step with the input pipeline in the loop, rocprofv3 kernel table + PMC passes
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r3m
mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_graph.py tests/test_gpu_kernels.py -x -q -s -m gpu -k "filip-FILIP or deferred" ) > $O/t1.log 2>&1
echo "t1 rc=$?" >> $O/t1.log
grep -E "FILIP gradient|passed|failed|rc=" $O/t1.log
for i in 1 2; do
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-loss-delta --no-roofline > $O/bench_resident_$i.log 2>&1
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-loss-delta --no-roofline --graph 0 > $O/bench_resident_eager_$i.log 2>&1
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-loss-delta --no-roofline --pipeline 1 > $O/bench_pipeline_$i.log 2>&1
done
for f in $O/bench_*.log; do echo $f; tail -1 $f | cut -c1-120; done
bash tools/profile_step.sh > $O/profile.log 2>&1
cp gpurun_out/prof/stats.txt $O/stats.txt 2>/dev/null
cp gpurun_out/prof/pmc_summary.txt $O/pmc_summary.txt 2>/dev/null
head -30 $O/stats.txt
tail -25 $O/pmc_summary.txt

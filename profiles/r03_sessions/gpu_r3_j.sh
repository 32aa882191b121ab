#!/bin/bash
# clocks / power of the chip while the CLIP step runs (is the step power-limited?)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r3j
mkdir -p $O
rocm-smi --showclocks --showpower --showmaxpower > $O/smi_idle.txt 2>&1
( timeout 300 python bench.py --steps 400 --warmup 5 --no-cpu-baseline --no-loss-delta --no-roofline > $O/bench.log 2>&1 ) &
BP=$!
sleep 25
for i in $(seq 1 12); do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power|power" >> $O/smi_run.txt
  echo "---" >> $O/smi_run.txt
  sleep 0.5
done
wait $BP
tail -1 $O/bench.log | cut -c1-200
cat $O/smi_idle.txt | grep -E "sclk|Power|power|Max" | head
echo ==== running
cat $O/smi_run.txt | head -60

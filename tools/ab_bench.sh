#!/bin/bash
# One parameterised A/B runner for bench.py (replaces the per-session scripts of rounds 1-3):
#   tools/ab_bench.sh OUTDIR "name1:ENV_A=1 ENV_B=2" "name2:" ... [-- extra bench.py arguments]
# runs `python bench.py --no-cpu-baseline --no-loss-delta --no-roofline --steps 30 <extra>` once per variant with the variant's
# environment, interleaved in the order given (repeat a name to get a second sample), writes OUTDIR/<name>_<i>.json and prints a table.
# On the GPU box: gpurun -- 'bash tools/ab_bench.sh gpurun_out/ab "base:" "x:DH_X=1" "base:" "x:DH_X=1"'
set -u
cd "$(dirname "$0")/.."
OUT=$1; shift
mkdir -p "$OUT"
EXTRA=()
VARS=()
while [ $# -gt 0 ]; do
  if [ "$1" == "--" ]; then shift; EXTRA=("$@"); break; fi
  VARS+=("$1"); shift
done
i=0
for v in "${VARS[@]}"; do
  name=${v%%:*}; envs=${v#*:}
  f="$OUT/${name}_$i"
  # shellcheck disable=SC2086
  env $envs timeout 600 python bench.py --no-cpu-baseline --no-loss-delta --no-roofline --steps 30 "${EXTRA[@]}" > "$f.json" 2> "$f.err"
  python - "$f.json" "$name" <<'PY' || tail -3 "$f.err"
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-24s %10.1f pairs/s %8.3f ms/step  graph %s graphs %s" % (sys.argv[2], d["value"], d["ms_per_step"], d["config"].get("step_graph"), d["config"].get("graphs_captured")))
PY
  i=$((i + 1))
done

"""Which ingredient of the step breaks hipGraph capture?  Each configuration of tools/diag_graph_child.py in its own process."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ENV = dict(DH_TOWER_STREAMS="0", DH_TEXT_PACKED="0", DH_POOLED_LAST="0")
for what in sys.argv[1:] or ["v1_fp32", "v2_fp32", "v6_fp32", "v3_step_fp32", "v5_step_fp32", "step_fp32"]:
    p = subprocess.run([sys.executable, os.path.join(HERE, "diag_graph_child.py"), what], env=dict(os.environ, **ENV), capture_output=True, text=True, timeout=300)
    res = [l for l in p.stdout.splitlines() if l.startswith("RESULT")]
    tail = " | ".join((p.stderr.strip().splitlines() or [""])[-2:])[:300]
    print("%-22s rc=%4d %s %s" % (what, p.returncode, res[0] if res else "", "" if res else tail), flush=True)

"""The `cpu_baseline` leg of bench.py run where the UNMODIFIED reference is importable (the build container): kind "reference".
    python tools/cpu_baseline_reference.py > profiles/r03_cpu_baseline_reference_container.json
The GPU box has no /root/reference, so BENCH lines there carry kind "port" (oracle/restated.py); this file is the reference-kind
number beside it, with the host it was taken on."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

out = bench.cpu_baseline()
out["where"] = "build container (no GPU); the MI355X box's host is 2 x EPYC 9575F, 128 cores"
print(json.dumps(out, indent=1))

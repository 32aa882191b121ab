"""Register / spill table of the kernels of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage), as the spill audits of
DESIGN_HISTORY.md s4 use it:   python tools/kernel_resources.py declip_amd/csrc/gemm_v4.hip [substring of the kernel name]"""
import re
import subprocess
import sys

src = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-ffast-math",
       "-fno-finite-math-only", "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur, rows = None, {}
for line in out.splitlines():
    m = re.search(r"remark: (?:Function Name: (\S+)|\s*([A-Za-z ]+?)(?: \[bytes/lane\])?: (\d+))", line)
    if not m:
        continue
    if m.group(1):
        cur = m.group(1)
        rows[cur] = {}
    elif cur:
        rows[cur][m.group(2).strip()] = int(m.group(3))
print("%-72s %5s %5s %6s %6s %7s" % ("kernel", "VGPR", "AGPR", "sSpill", "vSpill", "scratch"))
for k, v in rows.items():
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip() or k
    if pat in name:
        print("%-72s %5d %5d %6d %6d %7d" % (name[:72], v.get("VGPRs", -1), v.get("AGPRs", -1), v.get("SGPRs Spill", -1), v.get("VGPRs Spill", -1), v.get("ScratchSize", -1)))

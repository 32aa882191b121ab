"""Static check of the ONE assumption the hand-scheduled kernels make about the compiler: a register that an inline-asm LDS read
is still filling must not be touched before the wait that covers it.

gemm_v4 issues its transpose reads (`ds_read_b64_tr_b16`) as inline asm with plain "=v" outputs and waits for them much later
(`s_waitcnt lgkmcnt(0)` at the end of the load segment): the compiler does not know that the outputs are not valid yet.  If it
copies, spills or re-uses such a register before the wait, the kernel computes garbage from time to time -- the same class of bug
as the scheduler fetch of the dynamic tile distribution that round 3 found on the hardware (DESIGN_HISTORY.md s4).  This tool
disassembles the gfx950 code object inside a built object file and scans every kernel linearly:

    ds_read_b64_tr_b16 v[a:b], ...     -> a..b are PENDING
    s_waitcnt ... lgkmcnt(0)           -> nothing is pending any more
    any other instruction naming a pending register (as source or destination) -> VIOLATION

Usage: python tools/check_isa_async.py [declip_amd/csrc/gemm_v4.o]      (exit status 1 on violations)
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def disassemble(obj):
    tmp = tempfile.mkdtemp(prefix="dh_isa_")
    try:
        local = os.path.join(tmp, os.path.basename(obj))
        shutil.copy(obj, local)
        subprocess.run([OBJDUMP, "--offloading", local], capture_output=True, text=True, check=True, cwd=tmp)
        cos = [f for f in os.listdir(tmp) if "gfx950" in f]
        if not cos:
            raise RuntimeError("no gfx950 code object inside %s" % obj)
        return subprocess.run([OBJDUMP, "-d", os.path.join(tmp, cos[0])], capture_output=True, text=True, check=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def check(asm_text, only=None):
    """-> (per-kernel count of asynchronous reads, list of violations)"""
    counts, violations = {}, []
    kernel, pending = None, {}
    for raw in asm_text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", raw)
        if m:
            kernel, pending = m.group(1), {}
            continue
        if kernel is None or (only and only not in kernel):
            continue
        line = raw.split("//")[0].strip()
        if not line or line.startswith("."):
            continue
        parts = line.split(None, 1)
        op, args = parts[0], (parts[1] if len(parts) > 1 else "")
        if op == "s_endpgm":
            kernel = None
            continue
        if op == "ds_read_b64_tr_b16":
            dst = args.split(",")[0]
            touched = regs_of(args.split(",", 1)[1]) & set(pending)          # its ADDRESS register must not be pending either
            if touched:
                violations.append((kernel, line, sorted(touched)))
            for r in regs_of(dst):
                pending[r] = line
            counts[kernel] = counts.get(kernel, 0) + 1
            continue
        if op == "s_waitcnt":
            if re.search(r"lgkmcnt\(0\)", args):
                pending = {}
            continue
        if pending:
            hit = regs_of(args) & set(pending)
            if hit:
                violations.append((kernel, line, sorted(hit)))
    return counts, violations


def main():
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    obj = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "declip_amd", "csrc", "gemm_v4.o")
    counts, violations = check(disassemble(obj))
    for k, n in sorted(counts.items()):
        print("%-90s %4d asynchronous transpose reads" % (k[:90], n))
    for k, line, regs in violations:
        print("VIOLATION in %s: `%s` touches v%s while a transpose read into it is in flight" % (k, line, regs))
    print("%d kernels with asynchronous reads, %d violations" % (len(counts), len(violations)))
    return 1 if violations else 0


if __name__ == "__main__":
    sys.exit(main())

"""Instruction-stream map of one kernel of a built object (tuning aid): one character per instruction --
M mfma, D LDS-DMA, | s_barrier, s / l scratch store / load (spills), L / P buffer load / store, G global store, b branch.
    python tools/isa_map.py declip_amd/csrc/gemm_v4.o Lb0ELb0ELi3E"""
import re
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import check_isa_async as c  # noqa: E402

asm = c.disassemble(sys.argv[1])
k, ker = None, {}
for l in asm.splitlines():
    m = re.match(r"^[0-9a-f]+ <(\S+)>:", l)
    if m:
        k = m.group(1)
        ker[k] = []
    elif k:
        ker[k].append(l)
for name, lines in ker.items():
    if sys.argv[2] not in name:
        continue
    seq = []
    for l in lines:
        op = l.split()[0] if l.split() else ""
        ch = "."
        for pre, c_ in (("scratch_store", "s"), ("scratch_load", "l"), ("buffer_load", "L"), ("buffer_store", "P"), ("v_mfma", "M"),
                        ("s_barrier", "|"), ("global_store", "G"), ("global_load_lds", "D"), ("global_load", "g"), ("s_cbranch", "b"), ("s_branch", "b")):
            if op.startswith(pre):
                ch = c_
                break
        seq.append(ch)
    s = "".join(seq)
    print(name, len(lines))
    for i in range(0, len(s), 200):
        print("%5d %s" % (i, s[i:i + 200]))

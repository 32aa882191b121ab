"""Build libdeclip_hip.so in-tree with hipcc for gfx950 (no JIT cache: the .so travels with the repo).

    python -m declip_amd.build [--force]
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdeclip_hip.so")
STAMP = os.path.join(HERE, "csrc", ".build_stamp")
BASE_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-result", "-Wno-unused-value"]
FAST_MATH = ["-ffast-math", "-fno-finite-math-only"]
# Kernels that PRODUCE INDICES (FILIP's top-16 token selection and arg-max bytes, the nearest-neighbour search of the NN bank) are
# compiled with IEEE arithmetic: a reassociated sum or an approximate reciprocal in a comparison changes WHICH index wins a
# near-tie, and parity with the reference is judged on exactly those choices (one hardware failure of round 2 was a fast-math
# reciprocal).  Everything else keeps fast-math (contraction / reassociation inside fp32 epilogues, checked against the oracle
# at 1e-3).
IEEE_SOURCES = {"filip.hip", "declip_ops.hip"}
FLAGS = BASE_FLAGS + FAST_MATH                 # (the digest below covers both variants)


def flags_for(src):
    return BASE_FLAGS + ([] if os.path.basename(src) in IEEE_SOURCES else FAST_MATH)


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _digest():
    h = hashlib.sha256((" ".join(FLAGS) + "|ieee:" + ",".join(sorted(IEEE_SOURCES))).encode())
    files = sources() + [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".h")]
    files.append(os.path.join(os.path.dirname(HERE), "include", "declip_hip.h"))
    for f in files:
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(STAMP) and open(STAMP).read().strip() == dig:
        return LIB
    objs = []
    procs = []
    for src in sources():
        obj = src[:-4] + ".o"
        objs.append(obj)
        cmd = [hipcc] + flags_for(src) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on %s" % src)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--no-undefined", "-o", LIB] + objs   # a missing symbol fails the build, not the dlopen
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(STAMP, "w") as f:
        f.write(dig)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print("built", LIB)

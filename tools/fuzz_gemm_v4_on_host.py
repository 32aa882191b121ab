"""Random campaign over gemm_v4 on the host emulation (tests/hipemu, V4_EMU build): random tile counts (1 .. 4 items per emulated
workgroup), odd / even K-tile counts from 2 up, every epilogue flavour and operand layout, split-K and grouped weight gradients,
static / dynamic tile distribution, K-sliced schedules on / off -- each against an fp64 product.  Not part of the suite (seconds per
case); run it after touching the kernel's schedule or hand-over and before a GPU is available.

    python tools/fuzz_gemm_v4_on_host.py [seed] [seconds]
"""
import os
import random
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
bf = torch.bfloat16


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def rel_err(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def qgelu(x):
    return x * torch.sigmoid(1.702 * x)


def qgelu_grad(x):
    s = torch.sigmoid(1.702 * x)
    return s * (1 + 1.702 * x * (1 - s))


def case(ops, rng):
    from declip_amd.lib import EPI_DGELU, EPI_GELU
    kind = rng.choice(["bias", "gelu", "res", "dx", "dgelu", "dw", "group"])
    dyn, tail = rng.choice(["0", "1"]), rng.choice(["1", "2"])
    os.environ["DH_V4_DYNAMIC"], os.environ["DH_V4_TAIL"] = dyn, tail
    M, N = 256 * rng.randint(1, 6), 256 * rng.randint(1, 4)
    K = 64 * rng.randint(2, 9)
    if rng.random() < 0.15:
        K = 64 * rng.choice([24, 26])                       # long enough for the sliced tail of long tile lists
        M, N = 256 * rng.choice([3, 9]), 256 * rng.choice([1, 3])
    use_ws = rng.random() < 0.6
    ws = torch.empty((32 << 20) // 4, dtype=torch.float32) if use_ws else None
    desc = dict(kind=kind, M=M, N=N, K=K, dyn=dyn, tail=tail, ws=use_ws)
    sd = rng.randint(0, 1 << 20)
    tol = 1.2e-2
    if kind in ("bias", "gelu", "res"):
        A, B, bias = rnd(M, K, seed=sd).to(bf), rnd(N, K, seed=sd + 1, scale=0.2).to(bf), rnd(N, seed=sd + 2)
        pre = A.double() @ B.double().t() + bias.double()
        if kind == "bias":
            out = ops.gemm(A, B, bias=bias, ws=ws, force_generic=4)
            return desc, rel_err(out, pre), tol
        if kind == "res":
            R = rnd(M, N, seed=sd + 3).to(bf)
            out = ops.gemm(A, B, bias=bias, residual=R, ws=ws, force_generic=4)
            return desc, rel_err(out, pre + R.double()), tol
        aux = torch.empty(M, N, dtype=bf)
        out = ops.gemm(A, B, bias=bias, epilogue=EPI_GELU, aux=aux, force_generic=4)
        return desc, max(rel_err(aux, pre), rel_err(out, qgelu(pre))), tol
    if kind in ("dx", "dgelu"):
        dY, W = rnd(M, K, seed=sd).to(bf), rnd(K, N, seed=sd + 1, scale=0.1).to(bf)
        ref = dY.double() @ W.double()
        if kind == "dx":
            return desc, rel_err(ops.gemm(dY, W, b_kmajor=True, ws=ws, force_generic=4), ref), tol
        U = rnd(M, N, seed=sd + 2).to(bf)
        return desc, rel_err(ops.gemm(dY, W, b_kmajor=True, epilogue=EPI_DGELU, aux=U, force_generic=4), ref * qgelu_grad(U.double())), tol
    rows = 64 * rng.randint(8, 24)
    if kind == "dw":
        out_f, in_f = 256 * rng.randint(1, 2), 256 * rng.randint(1, 2)
        dY, X = rnd(rows, out_f, seed=sd).to(bf), rnd(rows, in_f, seed=sd + 1).to(bf)
        G0 = rnd(out_f, in_f, seed=sd + 2)
        gw, gb = G0.clone(), torch.ones(out_f)
        ops.gemm(dY, X, a_kmajor=True, b_kmajor=True, out=gw, accumulate=True, split_k=rng.choice([1, 2, 4, 8]), a_colsum=gb, ws=ws, force_generic=4)
        desc.update(rows=rows, out_f=out_f, in_f=in_f)
        return desc, max(rel_err(gw, G0.double() + dY.double().t() @ X.double()), rel_err(gb, 1 + dY.double().sum(0))), 2e-4
    n = rng.randint(2, 4)
    probs, refs = [], []
    for i in range(n):
        o_f, i_f = 256 * rng.randint(1, 2), 256 * rng.randint(1, 2)
        dy, x = rnd(rows, o_f, seed=sd + 10 * i).to(bf), rnd(rows, i_f, seed=sd + 10 * i + 1).to(bf)
        gw0, gb0 = rnd(o_f, i_f, seed=sd + 10 * i + 2), rnd(o_f, seed=sd + 10 * i + 3)
        with_b = rng.random() < 0.7
        probs.append((dy, x, gw0.clone(), gb0.clone() if with_b else None))
        refs.append((gw0.double() + dy.double().t() @ x.double(), gb0.double() + dy.double().sum(0) if with_b else None))
    ops.gemm_dw_group(probs, ws=torch.empty((64 << 20) // 4, dtype=torch.float32))
    err = 0.0
    for (dy, x, gw, gb), (rw, rb) in zip(probs, refs):
        err = max(err, rel_err(gw, rw), rel_err(gb, rb) if gb is not None else 0.0)
    desc.update(rows=rows, n=n)
    return desc, err, 2e-4


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    budget = float(sys.argv[2]) if len(sys.argv) > 2 else 300.0
    from hipemu_util import V4_SOURCES, emulated_gpu
    rng = random.Random(seed)
    t0, n, bad = time.time(), 0, 0
    with emulated_gpu(V4_SOURCES) as ops:
        while time.time() - t0 < budget:
            try:
                desc, err, tol = case(ops, rng)
            except Exception as e:                       # a refusal of the problem is fine (force_generic=4 raises), anything else is a finding
                msg = str(e)
                if "does not support" in msg.lower() or "unsupported" in msg.lower():
                    continue
                print("EXCEPTION", repr(e), flush=True)
                bad += 1
                continue
            n += 1
            if not (err < tol):
                bad += 1
                print("FAIL", desc, "err %.3e (tol %.1e)" % (err, tol), flush=True)
            elif n % 10 == 0:
                print("%d cases, %d findings, %.0f s" % (n, bad, time.time() - t0), flush=True)
    print("done: %d cases, %d findings" % (n, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

"""gemm_v4 -- the persistent 256 x 256 kernel that bench.py times -- on the host emulation (tests/hipemu; TEST INFRASTRUCTURE).

The kernel keeps its inline ISA (LDS-DMA, transpose reads, counted waits, kernarg-segment reads) behind a handful of macros, so
the SAME kernel body compiles as plain C++: an LDS-DMA is a synchronous 16-byte copy per lane, waits are no-ops, barriers are
counted over all 512 fibers of a workgroup, the emulated chip has 8 compute units (one workgroup per XCD list).  What this
pins before any GPU run: tile / K-tile indexing of every operand layout, the ring-buffer parity of the continuous K-tile stream
across the items of a workgroup (odd and even K-tile counts, 1 ... 3 items per workgroup), the hand-over to the next item inside
the last two K-tiles, barrier counts of the two wave groups (a mismatch deadlocks the emulation), every epilogue flavour, split-K
through the workspace, the grouped weight gradients, the K-sliced few-tile schedule, the max-sim and cross-entropy epilogues.
What it cannot see: anything the counted waits protect (a DMA lands "immediately" here)."""
import os

import pytest
import torch

from hipemu_util import V4_SOURCES, emulated_gpu
from test_gpu_gemm_v4 import TOL, quick_gelu, quick_gelu_grad, rel_err, rnd

bf = torch.bfloat16
# the default CPU suite keeps one case per mechanism (seconds each on the emulation); HIPEMU_SLOW=1 adds the rest (~3 more minutes)
SLOW = pytest.mark.skipif(os.environ.get("HIPEMU_SLOW") != "1", reason="tens of seconds on the host emulation: set HIPEMU_SLOW=1")


@pytest.fixture(scope="module")
def ops():
    with emulated_gpu(V4_SOURCES) as o:
        yield o


class _env:
    """Set environment switches the library reads per call (DH_V4_TAIL, DH_V4_TAIL_MINK, ...) around one block."""

    def __init__(self, **kv):
        self.kv = {k: str(v) for k, v in kv.items()}

    def __enter__(self):
        self.prev = {k: os.environ.get(k) for k in self.kv}
        os.environ.update(self.kv)
        self.prev_dyn = None
        if "DH_V4_DYNAMIC" in self.kv:          # (the library reads this switch once: set it through the C entry point)
            from declip_amd import ops as _o
            self.prev_dyn = _o.set_v4_dynamic(int(self.kv["DH_V4_DYNAMIC"]))

    def __exit__(self, *exc):
        if self.prev_dyn is not None:
            from declip_amd import ops as _o
            _o.set_v4_dynamic(self.prev_dyn)
        for k, v in self.prev.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _took_v4(ops, n=1):
    st = ops.gemm_stats()
    assert st["v4"] == n and sum(st.values()) == n, st


# 1 item per workgroup; 24 items = 3 per workgroup with nk = 3 (the ring-buffer parity flips from item to item), 10 items on 8
# workgroups with nk = 4, 10 with nk = 2 (the hand-over to the next item starts in the first K-tile), 9 with nk = 5
@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (1536, 1024, 192), pytest.param(1280, 512, 256, marks=SLOW), (2560, 256, 128),
                                   pytest.param(768, 768, 320, marks=SLOW)])
def test_v4_forward_bias_emulated(ops, M, N, K):
    A, B, bias = rnd(M, K, seed=1).to(bf), rnd(N, K, seed=2, scale=0.2).to(bf), rnd(N, seed=3)
    ops.gemm_stats(reset=True)
    out = ops.gemm(A, B, bias=bias, force_generic=4)
    _took_v4(ops)
    assert rel_err(out, A.double() @ B.double().t() + bias.double()) < TOL


@pytest.mark.parametrize("M,N,K", [(1280, 512, 192)])
def test_v4_forward_gelu_and_residual_emulated(ops, M, N, K):
    from declip_amd.lib import EPI_GELU
    A, B, bias = rnd(M, K, seed=4).to(bf), rnd(N, K, seed=5, scale=0.1).to(bf), rnd(N, seed=6)
    R = rnd(M, N, seed=7).to(bf)
    pre = A.double() @ B.double().t() + bias.double()
    aux = torch.empty(M, N, dtype=bf)
    out = ops.gemm(A, B, bias=bias, epilogue=EPI_GELU, aux=aux, force_generic=4)
    assert rel_err(aux, quick_gelu_grad(pre)) < TOL                  # aux = QuickGELU'(pre): the factor DH_EPI_DGELU multiplies by
    assert rel_err(out, quick_gelu(pre)) < TOL
    out = ops.gemm(A, B, bias=bias, residual=R, force_generic=4)
    assert rel_err(out, pre + R.double()) < TOL


@pytest.mark.parametrize("M,N,K", [pytest.param(1280, 512, 192, marks=SLOW), (768, 768, 128)])
def test_v4_dx_plain_and_dgelu_emulated(ops, M, N, K):
    from declip_amd.lib import EPI_DGELU
    dY, W = rnd(M, K, seed=8).to(bf), rnd(K, N, seed=9, scale=0.1).to(bf)     # W stored [out=K][in=N]: contraction-major B
    U = rnd(M, N, seed=10).to(bf)
    ref = dY.double() @ W.double()
    out = ops.gemm(dY, W, b_kmajor=True, force_generic=4)
    assert rel_err(out, ref) < TOL
    out = ops.gemm(dY, W, b_kmajor=True, epilogue=EPI_DGELU, aux=U, force_generic=4)
    assert rel_err(out, ref * U.double()) < TOL


@pytest.mark.parametrize("rows,out_f,in_f,use_ws", [(1024, 256, 512, True), (2048, 256, 256, False), pytest.param(1024, 256, 512, False, marks=SLOW),
                                                    pytest.param(2048, 256, 256, True, marks=SLOW)])
def test_v4_weight_grad_splitk_and_bias_grad_emulated(ops, rows, out_f, in_f, use_ws):
    dY, X = rnd(rows, out_f, seed=11).to(bf), rnd(rows, in_f, seed=12).to(bf)
    G0 = rnd(out_f, in_f, seed=13)
    gw, gb = G0.clone(), torch.ones(out_f)
    ws = torch.empty((64 << 20) // 4, dtype=torch.float32) if use_ws else None
    ops.gemm(dY, X, a_kmajor=True, b_kmajor=True, out=gw, accumulate=True, split_k=8, a_colsum=gb, ws=ws, force_generic=4)
    assert rel_err(gw, G0.double() + dY.double().t() @ X.double()) < 2e-4
    assert rel_err(gb, 1 + dY.double().sum(0)) < 2e-4


@pytest.mark.parametrize("residual", [False, True])
def test_v4_k_sliced_few_tile_schedule_emulated(ops, residual):
    """2 tiles on 8 compute units, 8 K-tiles: every tile is cut in K over the chip (private fp32 slice tiles + the fix-up kernel)."""
    M, N, K = 256, 512, 512
    A, B, bias = rnd(M, K, seed=14).to(bf), rnd(N, K, seed=15, scale=0.05).to(bf), rnd(N, seed=16)
    R = rnd(M, N, seed=17).to(bf) if residual else None
    ws = torch.empty((16 << 20) // 4, dtype=torch.float32)
    sliced = ops.gemm(A, B, bias=bias, residual=R, ws=ws, force_generic=4)
    plain = ops.gemm(A, B, bias=bias, residual=R, force_generic=4)
    ref = A.double() @ B.double().t() + bias.double() + (R.double() if residual else 0)
    assert rel_err(sliced, ref) < TOL and rel_err(plain, ref) < TOL


@pytest.fixture
def dynamic_tiles():
    """DH_V4_DYNAMIC=1 for one test: what declip_amd.dist.initialize selects for every multi-GPU job (items after a workgroup's first
    one come from per-XCD atomic counters; the general hand-over, never the incremental one)."""
    with _env(DH_V4_DYNAMIC=1):
        yield


def test_v4_dynamic_tile_distribution_emulated(ops, dynamic_tiles):
    """Forward (10 items of 2 K-tiles: the emulation runs the workgroups one after the other, so the first one of every XCD list takes
    ALL items of its list from the counter), dX + dGELU and grouped weight gradients under the dynamic distribution; twice:
    the scheduler state must be back at zero after every launch."""
    from declip_amd.lib import EPI_DGELU
    for rep in range(2):
        A, B, bias = rnd(1280, 128, seed=1).to(bf), rnd(512, 128, seed=2, scale=0.2).to(bf), rnd(512, seed=3)
        ops.gemm_stats(reset=True)
        out = ops.gemm(A, B, bias=bias, force_generic=4)
        _took_v4(ops)
        assert rel_err(out, A.double() @ B.double().t() + bias.double()) < TOL
        dY, W, U = rnd(768, 128, seed=8).to(bf), rnd(128, 768, seed=9, scale=0.1).to(bf), rnd(768, 768, seed=10).to(bf)
        out = ops.gemm(dY, W, b_kmajor=True, epilogue=EPI_DGELU, aux=U, force_generic=4)
        assert rel_err(out, (dY.double() @ W.double()) * U.double()) < TOL
        K, shapes = 512, [(256, 512), (256, 256)]
        probs, refs = [], []
        for i, (M, N) in enumerate(shapes):
            dy, x = rnd(K, M, seed=10 + i).to(bf), rnd(K, N, seed=20 + i).to(bf)
            gw0, gb0 = rnd(M, N, seed=30 + i), rnd(M, seed=40 + i)
            probs.append((dy, x, gw0.clone(), gb0.clone()))
            refs.append((gw0.double() + dy.double().t() @ x.double(), gb0.double() + dy.double().sum(0)))
        ops.gemm_dw_group(probs, ws=torch.empty((64 << 20) // 4, dtype=torch.float32))
        for (dy, x, gw, gb), (rw, rb) in zip(probs, refs):
            assert rel_err(gw, rw) < 2e-5 and rel_err(gb, rb) < 2e-5


def test_v4_grouped_weight_gradients_emulated(ops):
    K, shapes = 1024, [(256, 512), (512, 256), (256, 256)]
    probs, refs = [], []
    for i, (M, N) in enumerate(shapes):
        dy, x = rnd(K, M, seed=10 + i).to(bf), rnd(K, N, seed=20 + i).to(bf)
        gw0, gb0 = rnd(M, N, seed=30 + i), rnd(M, seed=40 + i)
        with_bias = i != 1
        probs.append((dy, x, gw0.clone(), gb0.clone() if with_bias else None))
        refs.append((gw0.double() + dy.double().t() @ x.double(), gb0.double() + dy.double().sum(0) if with_bias else None))
    ops.gemm_stats(reset=True)
    ops.gemm_dw_group(probs, ws=torch.empty((64 << 20) // 4, dtype=torch.float32))
    _took_v4(ops, len(shapes))
    for (dy, x, gw, gb), (rw, rb) in zip(probs, refs):
        assert rel_err(gw, rw) < 2e-5
        if gb is not None:
            assert rel_err(gb, rb) < 2e-5


@pytest.mark.parametrize("name,args", [("test_ce_fused_forward_and_backward", (300, 1000, 128)),
                                       pytest.param("test_maxsim_fused_forward_and_chunked_backward", (16, 16, 49, 256), marks=SLOW),
                                       ("test_maxsim_fused_forward_and_chunked_backward", (5, 16, 25, 128))])
def test_v4_fused_epilogues_of_the_gpu_suite_emulated(ops, monkeypatch, name, args):
    """The masked-LM cross-entropy epilogues (MODE_CE_FWD / MODE_CE_BWD, ragged last vocabulary tile) and FILIP's max-sim epilogue
    (MODE_MAXSIM): the SAME test functions as on the GPU (tests/test_gpu_kernels.py), at their small sizes."""
    import test_gpu_kernels as T
    monkeypatch.setattr(T, "cuda", torch.device("cpu"))
    monkeypatch.setattr(ops, "ce_fused_ok", lambda rows, weight: True)        # (the product gates ask for device tensors)
    monkeypatch.setattr(ops, "maxsim_fused_ok", lambda *a, **k: True)
    getattr(T, name)(*args)               # (these entry points launch gemm_v4 directly or fail: there is no other kernel behind them)

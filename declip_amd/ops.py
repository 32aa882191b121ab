"""Tensor-level wrappers over the C-ABI (one Python function per entry point).

These only validate shapes, allocate outputs with torch (memory plumbing) and enqueue the
HIP kernel on torch's current stream.  No arithmetic happens in Python/PyTorch here.
"""
import ctypes

import torch

from . import lib as L
from . import lib as L_
from .lib import DH_BF16, DH_F32, EPI_DGELU, EPI_GELU, EPI_NONE, GemmArgs, NcePair, check, dt, ptr, stream


def _req(cond, what):
    """argument check in front of a raw-pointer hand-over to the library: a real exception (bare asserts vanish under python -O)"""
    if not cond:
        raise L.DeclipHipError("bad argument: %s" % (what,))


def _contig(t, name):
    if not t.is_contiguous():
        raise L.DeclipHipError("%s must be contiguous" % name)
    return t


def gemm(A, B, *, a_kmajor=False, b_kmajor=False, bias=None, epilogue=EPI_NONE, residual=None, aux=None,
         out=None, out_dtype=None, accumulate=False, split_k=1, alpha=1.0, force_generic=False, a_colsum=None,
         pad_ok=False, dims=None, ws=None):
    """C[M,N] = epi(alpha * sum_k A(m,k) B(n,k) + bias).  A: [M,K] (or [K,M] if a_kmajor);
    B: [N,K] (or [K,N] if b_kmajor).  See include/declip_hip.h."""
    lib = L.load()
    _req(A.dim() == 2 and B.dim() == 2 and A.dtype == B.dtype, 'A.dim() == 2 and B.dim() == 2 and A.dtype == B.dtype')
    M, K = (A.shape[1], A.shape[0]) if a_kmajor else (A.shape[0], A.shape[1])
    N, Kb = (B.shape[1], B.shape[0]) if b_kmajor else (B.shape[0], B.shape[1])
    if dims is not None:            # logical (M, N, K) smaller/larger than the buffers (padded layouts, pad_ok)
        M, N, K = dims
    else:
        _req(K == Kb, "contraction mismatch %d vs %d" % (K, Kb))
    _req((A.shape[1] == 1 or A.stride(1) == 1) and (B.shape[1] == 1 or B.stride(1) == 1), '(A.shape[1] == 1 or A.stride(1) == 1) and (B.shape[1] == 1 or B.stride(1) == 1)')  # (a 1-wide dim may report any stride)
    if out is None:
        out = torch.empty(M, N, device=A.device, dtype=out_dtype or A.dtype)
    _req((dims is not None or out.shape == (M, N)) and out.stride(1) == 1, '(dims is not None or out.shape == (M, N)) and out.stride(1) == 1')
    a = GemmArgs()
    a.dtype, a.c_dtype = dt(A), dt(out)
    a.a_kmajor, a.b_kmajor = int(a_kmajor), int(b_kmajor)
    a.M, a.N, a.K = M, N, K
    a.A, a.lda, a.B, a.ldb, a.C, a.ldc = ptr(A), max(A.stride(0), A.shape[1]), ptr(B), max(B.stride(0), B.shape[1]), ptr(out), out.stride(0)
    if bias is not None:
        _req(bias.dtype == torch.float32 and (dims is not None or bias.numel() == N), 'bias.dtype == torch.float32 and (dims is not None or bias.numel() == N)')
    a.bias = ptr(bias)
    a.epilogue = epilogue
    if residual is not None:
        _req(residual.dtype == out.dtype and residual.shape == out.shape, 'residual.dtype == out.dtype and residual.shape == out.shape')
        a.residual, a.ldr = ptr(residual), residual.stride(0)
    if aux is not None:
        _req(aux.shape == (M, N), 'aux.shape == (M, N)')
        _req(aux.dtype == (out.dtype if epilogue == EPI_GELU else A.dtype), 'aux.dtype == (out.dtype if epilogue == EPI_GELU else A.dtype)')
        a.aux, a.ldaux = ptr(aux), aux.stride(0)
    a.accumulate, a.split_k, a.alpha, a.force_generic = int(accumulate), int(split_k), float(alpha), int(force_generic)
    if a_colsum is not None:
        _req(a_kmajor and a_colsum.dtype == torch.float32 and a_colsum.numel() == M, 'a_kmajor and a_colsum.dtype == torch.float32 and a_colsum.numel() == M')
        a.a_colsum = ptr(a_colsum)
    a.pad_ok = int(pad_ok)
    if ws is not None:              # caller scratch for the split-K partial tiles (v4 kernel)
        a.ws, a.ws_bytes = ptr(ws), ws.numel() * ws.element_size()
    check(lib.dh_gemm(ctypes.byref(a), stream()), "dh_gemm")
    return out


def gemm_dw_group(problems, ws=None, first_touch=False):
    """problems: list of (dy [rows, out], x [rows, in], gw [out, in] fp32, gb [out] fp32 or None), all with the same `rows`:
    gw += dy^T x and gb += colsum(dy) for every problem in ONE launch of the persistent kernel (dh_gemm_group).
    first_touch: these are the first contributions of the step to gw / gb: they are WRITTEN (dh_gemm_args.accumulate = 2; the slots
    need not be zero and are not read)."""
    n = len(problems)
    arr = (GemmArgs * n)()
    for a, (dy, x, gw, gb) in zip(arr, problems):
        _req(dy.dim() == 2 and x.dim() == 2 and dy.shape[0] == x.shape[0] and dy.dtype == x.dtype, 'dy.dim() == 2 and x.dim() == 2 and dy.shape[0] == x.shape[0] and dy.dtype == x.dtype')
        _req(gw.dtype == torch.float32 and gw.shape == (dy.shape[1], x.shape[1]) and gw.stride(1) == 1, 'gw.dtype == torch.float32 and gw.shape == (dy.shape[1], x.shape[1]) and gw.stride(1) == 1')
        _req(dy.stride(1) == 1 and x.stride(1) == 1, 'dy.stride(1) == 1 and x.stride(1) == 1')
        a.dtype, a.c_dtype = dt(dy), DH_F32
        a.a_kmajor, a.b_kmajor = 1, 1
        a.M, a.N, a.K = dy.shape[1], x.shape[1], dy.shape[0]
        a.A, a.lda, a.B, a.ldb, a.C, a.ldc = ptr(dy), dy.stride(0), ptr(x), x.stride(0), ptr(gw), gw.stride(0)
        a.accumulate, a.alpha = (2 if first_touch else 1), 1.0
        a.split_k = max(1, min(1024 // max(((a.M + 127) // 128) * ((a.N + 127) // 128), 1), a.K // 512))   # used only on the one-by-one path
        if gb is not None:
            _req(gb.dtype == torch.float32 and gb.numel() == a.M, 'gb.dtype == torch.float32 and gb.numel() == a.M')
            a.a_colsum = ptr(gb)
        if ws is not None:
            a.ws, a.ws_bytes = ptr(ws), ws.numel() * ws.element_size()
    check(L.load().dh_gemm_group(arr, n, stream()), "dh_gemm_group")


def zero_ranges(base, table_dev, n, max_len):
    """base[lo:hi] = 0 for the n (lo, hi) rows of the int64 device table (dh_zero_ranges)."""
    _req(base.dtype == torch.float32 and table_dev.dtype == torch.int64 and table_dev.is_contiguous(), "zero_ranges: fp32 buffer, int64 [n, 2] table")
    check(L.load().dh_zero_ranges(ptr(base), ptr(table_dev), int(n), int(max_len), stream()), "dh_zero_ranges")


def gemm_stats(reset=False):
    """dh_gemm launches per kernel family since the last reset (test instrumentation, include/declip_hip.h)."""
    out = (ctypes.c_longlong * 5)()
    check(L.load().dh_gemm_stats(out, int(reset)), "dh_gemm_stats")
    return dict(v4=out[0], v3=out[1], glds=out[2], mfma_tiles=out[3], generic=out[4])


def set_v4_dynamic(mode):
    """Tile distribution of the persistent GEMM (0 static, 1 dynamic: include/declip_hip.h dh_gemm_v4_set_dynamic); returns the previous mode."""
    return int(L.load().dh_gemm_v4_set_dynamic(int(mode)))


def colsum(X, out, accumulate=True):
    _req(X.dim() == 2 and X.stride(1) == 1 and out.dtype == torch.float32 and out.numel() == X.shape[1], 'X.dim() == 2 and X.stride(1) == 1 and out.dtype == torch.float32 and out.numel() == X.shape[1]')
    check(L.load().dh_colsum(dt(X), ptr(X), X.stride(0), X.shape[0], X.shape[1], ptr(out), int(accumulate), stream()),
          "dh_colsum")
    return out


def layernorm_fwd(x, w, b, eps=1e-5, save_stats=True):
    _contig(x, "x")
    rows, d = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(rows, device=x.device, dtype=torch.float32) if save_stats else None
    rstd = torch.empty(rows, device=x.device, dtype=torch.float32) if save_stats else None
    check(L.load().dh_layernorm_fwd(dt(x), ptr(x), ptr(w), ptr(b), ptr(y), ptr(mean), ptr(rstd), rows, d, eps, stream()),
          "dh_layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(dy, x, w, mean, rstd, dw, db, dres=None):
    """dx = LN'(dy) (+ dres); dw/db (fp32) accumulated into."""
    _contig(dy, "dy"), _contig(x, "x")
    rows, d = x.shape
    lib = L.load()
    dx = torch.empty_like(x)
    nbytes = lib.dh_layernorm_bwd_ws_bytes(rows, d)
    ws = torch.empty(max(nbytes, 4) // 4, device=x.device, dtype=torch.float32)
    check(lib.dh_layernorm_bwd(dt(x), ptr(dy), ptr(x), ptr(w), ptr(mean), ptr(rstd), ptr(dres), ptr(dx), ptr(dw), ptr(db),
                               rows, d, ptr(ws), nbytes, stream()), "dh_layernorm_bwd")
    return dx


def layernorm_bwd_part(dy, x, w, mean, rstd, dw, db, part, dres=None):
    """layernorm_bwd with the weight / bias gradient reduction deferred: the per-block partials stay in `part` (fp32, at least
    layernorm_bwd_ws_elems(rows, d) elements, alive until ln_reduce_many ran).  Returns (dx, nb): nb partial rows, 0 = accumulated
    into dw / db directly."""
    _contig(dy, "dy"), _contig(x, "x")
    rows, d = x.shape
    lib = L.load()
    dx = torch.empty_like(x)
    nb = ctypes.c_int(0)
    _req(part.dtype == torch.float32 and part.is_contiguous(), "part: contiguous fp32")
    check(lib.dh_layernorm_bwd_part(dt(x), ptr(dy), ptr(x), ptr(w), ptr(mean), ptr(rstd), ptr(dres), ptr(dx), ptr(dw), ptr(db),
                                    rows, d, ptr(part), part.numel() * 4, ctypes.byref(nb), stream()), "dh_layernorm_bwd_part")
    return dx, int(nb.value)


def layernorm_bwd_ws_elems(rows, d):
    return max(int(L.load().dh_layernorm_bwd_ws_bytes(rows, d)), 4) // 4


def ln_reduce_many(items):
    """items: [(part, nb, d, dw, db)] from layernorm_bwd_part: dw += column sums of part[:nb, :d], db += those of part[:nb, d:2d], for
    all items in ONE launch per 32 (dh_ln_reduce_many)."""
    items = [it for it in items if it[1] > 0]
    if not items:
        return
    arr = (L.LnPart * len(items))()
    for a, (part, nb, d, dw, db) in zip(arr, items):
        _req(dw.dtype == torch.float32 and db.dtype == torch.float32 and dw.numel() == d and db.numel() == d, "ln_reduce_many: dw / db fp32 [d]")
        a.part, a.dw, a.db, a.nb, a.d = ptr(part), ptr(dw), ptr(db), nb, d
    check(L.load().dh_ln_reduce_many(arr, len(items), stream()), "dh_ln_reduce_many")


def block_native_available():
    """True: dh_block_fwd / dh_block_bwd can take this process's tensors (the torch-CPU stand-ins of the tests replace this by False)."""
    return True


_BLOCK_OFFS = {}


def block_act_layout(dtype, rows, d, heads, b, L):
    """(total bytes, {name: byte offset}) of a block's activation slab (dh_block_act_bytes / dh_block_act_offsets)."""
    key = (dtype, rows, d, heads, b, L)
    lay = _BLOCK_OFFS.get(key)
    if lay is None:
        lib = L_.load()
        out = (ctypes.c_int64 * 12)()
        check(lib.dh_block_act_offsets(dtype, rows, d, heads, b, L, out), "dh_block_act_offsets")
        names = ("h1", "qkv", "a", "x_mid", "h2", "u", "g", "mean1", "rstd1", "mean2", "rstd2", "lse")
        lay = (int(lib.dh_block_act_bytes(dtype, rows, d, heads, b, L)), dict(zip(names, (int(v) for v in out))))
        if len(_BLOCK_OFFS) > 256:
            _BLOCK_OFFS.clear()
        _BLOCK_OFFS[key] = lay
    return lay


def block_bwd_scratch_bytes(dtype, rows, d):
    return int(L_.load().dh_block_bwd_scratch_bytes(dtype, rows, d))


def block_fwd(args):
    """One ResidualAttentionBlock forward from ONE call (dh_block_fwd; args: lib.BlockArgs filled by the engine)."""
    check(L_.load().dh_block_fwd(ctypes.byref(args), stream()), "dh_block_fwd")


def block_bwd(args):
    """Its backward, incl. the block's grouped weight gradients (dh_block_bwd); writes args.ln_nb1 / ln_nb2."""
    check(L_.load().dh_block_bwd(ctypes.byref(args), stream()), "dh_block_bwd")


def attn_fwd(qkv, b, Lq, heads, causal):
    _contig(qkv, "qkv")
    d3 = qkv.shape[-1]
    d = d3 // 3
    hd = d // heads
    out = torch.empty(b * Lq, d, device=qkv.device, dtype=qkv.dtype)
    lse = torch.empty(b, heads, Lq, device=qkv.device, dtype=torch.float32)
    check(L.load().dh_attn_fwd(dt(qkv), ptr(qkv), ptr(out), ptr(lse), b, Lq, heads, hd, int(causal), stream()), "dh_attn_fwd")
    return out, lse


def attn_bwd(qkv, out, dout, lse, b, Lq, heads, causal):
    _contig(qkv, "qkv"), _contig(out, "out"), _contig(dout, "dout")
    d = qkv.shape[-1] // 3
    hd = d // heads
    dqkv = torch.empty_like(qkv)
    check(L.load().dh_attn_bwd(dt(qkv), ptr(qkv), ptr(out), ptr(dout), ptr(lse), ptr(dqkv), b, Lq, heads, hd, int(causal),
                               stream()), "dh_attn_bwd")
    return dqkv


def attn_varlen_fwd(qkv, cu, rows, b, Lmax, heads, causal):
    """attention on packed sequences: qkv [rows_pad, 3d], sequence i = rows cu[i] .. cu[i+1]; out [rows_pad, d] (tail rows zero)."""
    _contig(qkv, "qkv")
    _req(cu.dtype == torch.int32 and cu.numel() == b + 1, 'cu.dtype == torch.int32 and cu.numel() == b + 1')
    d = qkv.shape[-1] // 3
    out = torch.empty(qkv.shape[0], d, device=qkv.device, dtype=qkv.dtype)      # the tail rows [rows, rows_pad) are zeroed by the call
    lse = torch.empty(b, heads, Lmax, device=qkv.device, dtype=torch.float32)
    check(L.load().dh_attn_varlen_fwd(dt(qkv), ptr(qkv), ptr(out), ptr(lse), ptr(_contig(cu, "cu")), b, Lmax, heads, d // heads, int(causal),
                                      int(rows), int(qkv.shape[0]), stream()), "dh_attn_varlen_fwd")
    return out, lse


def attn_varlen_bwd(qkv, out, dout, lse, cu, rows, b, Lmax, heads, causal):
    _contig(qkv, "qkv"), _contig(out, "out"), _contig(dout, "dout")
    d = qkv.shape[-1] // 3
    dqkv = torch.empty_like(qkv)                                                 # (tail rows zeroed by the call)
    check(L.load().dh_attn_varlen_bwd(dt(qkv), ptr(qkv), ptr(out), ptr(dout), ptr(lse), ptr(dqkv), ptr(_contig(cu, "cu")), b, Lmax, heads,
                                      d // heads, int(causal), int(rows), int(qkv.shape[0]), stream()), "dh_attn_varlen_bwd")
    return dqkv


def attn_bucketed_fwd(qkv, cu, order, ranges, rows, b, Lmax, L_short, heads, causal):
    """attn_varlen_fwd in two length buckets (dh_attn_bucketed_fwd): order int32 [b] (short sequences first), ranges int32 [4]."""
    _contig(qkv, "qkv")
    _req(cu.dtype == torch.int32 and order.dtype == torch.int32 and ranges.dtype == torch.int32 and order.numel() == b and ranges.numel() == 4,
         "attn_bucketed_fwd: cu / order / ranges int32")
    d = qkv.shape[-1] // 3
    out = torch.empty(qkv.shape[0], d, device=qkv.device, dtype=qkv.dtype)
    lse = torch.empty(b, heads, Lmax, device=qkv.device, dtype=torch.float32)
    check(L.load().dh_attn_bucketed_fwd(dt(qkv), ptr(qkv), ptr(out), ptr(lse), ptr(cu), ptr(order), ptr(ranges), b, Lmax, L_short, heads, d // heads,
                                        int(causal), int(rows), int(qkv.shape[0]), stream()), "dh_attn_bucketed_fwd")
    return out, lse


def attn_bucketed_bwd(qkv, out, dout, lse, cu, order, ranges, rows, b, Lmax, L_short, heads, causal):
    _contig(qkv, "qkv"), _contig(out, "out"), _contig(dout, "dout")
    d = qkv.shape[-1] // 3
    dqkv = torch.empty_like(qkv)
    check(L.load().dh_attn_bucketed_bwd(dt(qkv), ptr(qkv), ptr(out), ptr(dout), ptr(lse), ptr(dqkv), ptr(cu), ptr(order), ptr(ranges), b, Lmax,
                                        L_short, heads, d // heads, int(causal), int(rows), int(qkv.shape[0]), stream()), "dh_attn_bucketed_bwd")
    return dqkv


def attn_pooled_fwd(q, kv, row0, nkeys, heads, Lmax):
    """one query per sequence: q [b, d], kv [rows, 2d]; keys of sequence i = kv rows row0[i] .. row0[i] + nkeys[i] - 1."""
    _contig(q, "q"), _contig(kv, "kv")
    b, d = q.shape
    _req(kv.shape[1] == 2 * d and row0.dtype == torch.int32 and nkeys.dtype == torch.int32 and row0.numel() == b == nkeys.numel(), 'kv.shape[1] == 2 * d and row0.dtype == torch.int32 and nkeys.dtype == torch.int32 and row0.numel() == b == nkeys.numel()')
    out = torch.empty_like(q)
    lse = torch.empty(b, heads, device=q.device, dtype=torch.float32)
    check(L.load().dh_attn_pooled_fwd(dt(q), ptr(q), ptr(kv), ptr(out), ptr(lse), ptr(_contig(row0, "row0")), ptr(_contig(nkeys, "nkeys")), b,
                                      heads, d // heads, Lmax, stream()), "dh_attn_pooled_fwd")
    return out, lse


def attn_pooled_bwd(q, kv, dout, lse, row0, nkeys, heads, Lmax, ordered=False):
    """ordered: the sequences lie in row order (row0 ascending) -- the launch then zeroes the dkv rows no sequence owns itself, instead
    of a fill of the whole [rows, 2d] buffer before it (78 + 45 MB per CLIP step at b = 512)."""
    _contig(q, "q"), _contig(kv, "kv"), _contig(dout, "dout")
    b, d = q.shape
    dq = torch.empty_like(q)
    dkv = torch.empty_like(kv) if ordered else torch.zeros_like(kv)
    check(L.load().dh_attn_pooled_bwd(dt(q), ptr(q), ptr(kv), ptr(dout), ptr(lse), ptr(dq), ptr(dkv), ptr(row0), ptr(nkeys), b, heads,
                                      d // heads, Lmax, kv.shape[0] if ordered else 0, stream()), "dh_attn_pooled_bwd")
    return dq, dkv


def text_embed_fwd(ids, table, pos, dtype):
    b, Lq = ids.shape
    d = table.shape[1]
    x = torch.empty(b * Lq, d, device=table.device, dtype=dtype)
    check(L.load().dh_text_embed_fwd(dt(x), ptr(ids), ptr(table), ptr(pos), ptr(x), b, Lq, d, stream()), "dh_text_embed_fwd")
    return x


_EMBED_WS = {}


def embed_table_grad(ids, dx, dtable):
    """dtable [vocab, d] += sum over rows r of dx[r] at row ids[r]: sort-by-id segmented reduction (dh_embed_table_grad).  The scratch
    is cached per device and size (created in the eager warm-up steps, so a captured step replays with it)."""
    rows, d = ids.numel(), dx.shape[-1]
    _req(dx.numel() == rows * d and dtable.dim() == 2 and dtable.shape[1] == d and dtable.dtype == torch.float32 and ids.dtype == torch.int64,
         "embed_table_grad: ids [rows] int64, dx [rows, d], dtable [vocab, d] fp32")
    vocab = dtable.shape[0]
    need = L.load().dh_embed_table_grad_ws_bytes(rows, vocab)
    key = (str(dx.device), need)
    ws = _EMBED_WS.get(key)
    if ws is None:
        ws = _EMBED_WS[key] = torch.empty(need // 4, device=dx.device, dtype=torch.int32)
    check(L.load().dh_embed_table_grad(dt(dx), ptr(_contig(ids, "ids")), ptr(_contig(dx, "dx")), ptr(_contig(dtable, "dtable")), rows, d, vocab,
                                       ptr(ws), need, stream()), "dh_embed_table_grad")


def text_embed_bwd(ids, dx, dtable, dpos, hot_ids=(0,)):
    """Token table by the sorted segmented reduction (no per-element atomics; `hot_ids` is kept for callers of the scatter-add entry
    dh_text_embed_bwd and unused here), positions by the batch reduction."""
    b, Lq = ids.shape
    d = dx.shape[-1]
    if dtable is not None:
        embed_table_grad(ids, dx, dtable)
    if dpos is not None:
        hot = (ctypes.c_int64 * 1)(0)
        check(L.load().dh_text_embed_bwd(dt(dx), ptr(ids), ptr(dx), None, ptr(dpos), b, Lq, d, hot, 0, stream()), "dh_text_embed_bwd")


def text_embed_packed_fwd(ids_p, pos_idx, table, pos, dtype, rows, rows_pad):
    """packed captions: x [rows_pad, d], x[r] = table[ids_p[r]] + pos[pos_idx[r]] for r < rows, zero rows after."""
    _req(ids_p.dtype == torch.int64 and pos_idx.dtype == torch.int32 and ids_p.numel() >= rows and pos_idx.numel() >= rows, 'ids_p.dtype == torch.int64 and pos_idx.dtype == torch.int32 and ids_p.numel() >= rows and pos_idx.numel() >= rows')
    d = table.shape[1]
    x = torch.empty(rows_pad, d, device=table.device, dtype=dtype)
    check(L.load().dh_text_embed_packed_fwd(dt(x), ptr(_contig(ids_p, "ids")), ptr(_contig(pos_idx, "pos_idx")), ptr(table), ptr(pos), ptr(x),
                                            rows, rows_pad, d, stream()), "dh_text_embed_packed_fwd")
    return x


def text_embed_packed_bwd(ids_p, cu, dx, dtable, dpos, rows, Lmax, hot_ids=()):
    """gradients of the packed embedding: token table by scatter-add over the packed ids, positions by a per-position reduction."""
    d = dx.shape[1]
    if dtable is not None:
        embed_table_grad(ids_p[:rows], dx[:rows], dtable)
    if dpos is not None:
        _req(cu.dtype == torch.int32, 'cu.dtype == torch.int32')
        check(L.load().dh_packed_pos_grad(dt(dx), ptr(dx), ptr(_contig(cu, "cu")), cu.numel() - 1, Lmax, d, ptr(dpos), stream()),
              "dh_packed_pos_grad")


IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)     # transforms.Normalize of the reference pipelines


def image_prep_u8(src, out_hw, crop_xy=None, flip=None, mean=IMAGENET_MEAN, std=IMAGENET_STD, out=None, c0=0):
    """uint8 [b, Hs, Ws, 3] on the GPU -> fp32 [b, C, H, W] channels c0..c0+2 (crop window, optional mirror, normalise).
    crop_xy: int32 [b, 2] (x0, y0) device tensor or None; flip: uint8/bool [b] device tensor or None."""
    _req(src.dtype == torch.uint8 and src.dim() == 4 and src.shape[3] == 3 and src.is_contiguous(), 'src.dtype == torch.uint8 and src.dim() == 4 and src.shape[3] == 3 and src.is_contiguous()')
    b, Hs, Ws, _ = src.shape
    H, W = out_hw
    if out is None:
        out = torch.empty(b, 3, H, W, device=src.device, dtype=torch.float32)
    _req(out.dtype == torch.float32 and out.is_contiguous() and out.shape[0] == b and tuple(out.shape[2:]) == (H, W), 'out.dtype == torch.float32 and out.is_contiguous() and out.shape[0] == b and tuple(out.shape[2:]) == (H, W)')
    if crop_xy is not None:
        _req(crop_xy.dtype == torch.int32 and crop_xy.shape == (b, 2) and crop_xy.is_contiguous(), 'crop_xy.dtype == torch.int32 and crop_xy.shape == (b, 2) and crop_xy.is_contiguous()')
    if flip is not None:
        flip = flip.to(torch.uint8).contiguous()
        _req(flip.shape == (b,), 'flip.shape == (b,)')
    m3, s3 = (ctypes.c_float * 3)(*mean), (ctypes.c_float * 3)(*std)
    check(L.load().dh_image_prep_u8(ptr(src), b, Hs, Ws, ptr(crop_xy), ptr(flip), m3, s3, ptr(out), out.shape[1], c0, H, W, stream()),
          "dh_image_prep_u8")
    return out


def image_resized_crop_u8(src, params, out_hw, flip=None, mean=IMAGENET_MEAN, std=IMAGENET_STD, out=None, c0=0, round_u8=True):
    """uint8 canvas [b, Hs, Ws, 3] on the GPU -> fp32 [b, C, H, W] channels c0..c0+2: crop box -> antialiased bilinear resize -> window
    -> optional mirror -> ToTensor -> Normalize.  params: int32 [b, 8] device tensor (x0, y0, w, h, Wf, Hf, ox, oy), see
    include/declip_hip.h; flip: uint8/bool [b] device tensor or None."""
    _req(src.dtype == torch.uint8 and src.dim() == 4 and src.shape[3] == 3 and src.is_contiguous(), 'src.dtype == torch.uint8 and src.dim() == 4 and src.shape[3] == 3 and src.is_contiguous()')
    b, Hs, Ws, _ = src.shape
    H, W = out_hw
    _req(params.dtype == torch.int32 and params.shape == (b, 8) and params.is_contiguous(), 'params.dtype == torch.int32 and params.shape == (b, 8) and params.is_contiguous()')
    if out is None:
        out = torch.empty(b, 3, H, W, device=src.device, dtype=torch.float32)
    _req(out.dtype == torch.float32 and out.is_contiguous() and out.shape[0] == b and tuple(out.shape[2:]) == (H, W), 'out.dtype == torch.float32 and out.is_contiguous() and out.shape[0] == b and tuple(out.shape[2:]) == (H, W)')
    if flip is not None:
        flip = flip.to(torch.uint8).contiguous()
        _req(flip.shape == (b,), 'flip.shape == (b,)')
    m3, s3 = (ctypes.c_float * 3)(*mean), (ctypes.c_float * 3)(*std)
    check(L.load().dh_image_resized_crop_u8(ptr(src), b, Hs, Ws, ptr(params), ptr(flip), m3, s3, ptr(out), out.shape[1], c0, H, W,
                                            int(round_u8), stream()), "dh_image_resized_crop_u8")
    return out


def im2row(images, c0, patch, dtype, out=None):
    _contig(images, "images")
    _req(images.dtype == torch.float32, 'images.dtype == torch.float32')
    b, ctot, H, W = images.shape
    rows = out if out is not None else torch.empty(b * (H // patch) * (W // patch), 3 * patch * patch, device=images.device, dtype=dtype)
    _req(rows.is_contiguous() and rows.shape == (b * (H // patch) * (W // patch), 3 * patch * patch), 'rows.is_contiguous() and rows.shape == (b * (H // patch) * (W // patch), 3 * patch * patch)')
    check(L.load().dh_im2row(dt(rows), ptr(images), ctot, c0, ptr(rows), b, H, W, patch, stream()), "dh_im2row")
    return rows


def vit_assemble_fwd(patches, cls, pos, b, npatch):
    d = patches.shape[-1]
    x = torch.empty(b * (npatch + 1), d, device=patches.device, dtype=patches.dtype)
    check(L.load().dh_vit_assemble_fwd(dt(x), ptr(patches), ptr(cls), ptr(pos), ptr(x), b, npatch, d, stream()), "dh_vit_assemble_fwd")
    return x


def vit_assemble_bwd(dx, dcls, dpos, b, npatch):
    d = dx.shape[-1]
    check(L.load().dh_vit_assemble_bwd(dt(dx), ptr(dx), ptr(dcls), ptr(dpos), b, npatch, d, stream()), "dh_vit_assemble_bwd")


def pool_rows_fwd(x, idx, b, Lq):
    d = x.shape[-1]
    out = torch.empty(b, d, device=x.device, dtype=x.dtype)
    check(L.load().dh_pool_rows_fwd(dt(x), ptr(x), ptr(idx), ptr(out), b, Lq, d, stream()), "dh_pool_rows_fwd")
    return out


def pool_rows_bwd(dout, idx, b, Lq):
    d = dout.shape[-1]
    dx = torch.empty(b * Lq, d, device=dout.device, dtype=dout.dtype)
    check(L.load().dh_pool_rows_bwd(dt(dout), ptr(dout), ptr(idx), ptr(dx), b, Lq, d, stream()), "dh_pool_rows_bwd")
    return dx


def l2norm_fwd(x, eps):
    _contig(x, "x")
    rows, d = x.shape
    y = torch.empty(rows, d, device=x.device, dtype=torch.float32)
    norm = torch.empty(rows, device=x.device, dtype=torch.float32)
    check(L.load().dh_l2norm_fwd(dt(x), ptr(x), ptr(y), ptr(norm), rows, d, eps, stream()), "dh_l2norm_fwd")
    return y, norm


def l2norm_bwd(x, norm, dy, eps):
    rows, d = x.shape
    dx = torch.empty_like(x)
    check(L.load().dh_l2norm_bwd(dt(x), ptr(x), ptr(norm), ptr(_contig(dy, "dy")), ptr(dx), rows, d, eps, stream()), "dh_l2norm_bwd")
    return dx


def _pair_array(pairs):
    arr = (NcePair * len(pairs))()
    for i, p in enumerate(pairs):
        arr[i].Q, arr[i].K = ptr(p[0]), ptr(p[1])
        arr[i].dQ = ptr(p[2]) if len(p) > 2 else None
        arr[i].dK = ptr(p[3]) if len(p) > 3 else None
    return arr


def _int_array(vals, n):
    if vals is None:
        return None
    _req(len(vals) == n, 'len(vals) == n')
    return (ctypes.c_int * n)(*[int(v) for v in vals])


def infonce_fwd(pairs, scale, label0, want_logits=False, label0s=None, excl0s=None):
    """pairs: list of (Q[b,D], K[B,D]) fp32.  scale: 1-element fp32 device tensor.
    Returns row_loss, row_lse, correct1, correct5 ([P,b] fp32) and optional logits [P,b,B]."""
    Q0, K0 = pairs[0][0], pairs[0][1]
    b, D = Q0.shape
    B = K0.shape[0]
    for q, k in pairs:
        _req(q.dtype == torch.float32 and k.dtype == torch.float32 and q.shape == (b, D) and k.shape == (B, D), 'q.dtype == torch.float32 and k.dtype == torch.float32 and q.shape == (b, D) and k.shape == (B, D)')
        _contig(q, "Q"), _contig(k, "K")
    P = len(pairs)
    mk = lambda: torch.empty(P, b, device=Q0.device, dtype=torch.float32)
    row_loss, row_lse, c1, c5 = mk(), mk(), mk(), mk()
    logits = torch.empty(P, b, B, device=Q0.device, dtype=torch.float32) if want_logits else None
    arr = _pair_array(pairs)
    lib = L.load()
    nbytes = lib.dh_infonce_ws_bytes(P, b, B)
    ws = torch.empty(nbytes // 4, device=Q0.device, dtype=torch.float32)
    check(lib.dh_infonce_fwd(arr, P, b, B, D, ptr(scale), int(label0), _int_array(label0s, P), _int_array(excl0s, P),
                             ptr(row_loss), ptr(row_lse), ptr(c1), ptr(c5), ptr(logits), ptr(ws), nbytes, stream()), "dh_infonce_fwd")
    return row_loss, row_lse, c1, c5, logits


def infonce_bwd(pairs, scale, label0, row_lse, g_row, need=None, label0s=None, excl0s=None):
    """pairs: list of (Q, K); returns list of (dQ, dK) and dscale (1-element).  need: optional list of
    (need_dQ, need_dK) flags -- unneeded gradients are skipped (None returned)."""
    Q0, K0 = pairs[0]
    b, D = Q0.shape
    B = K0.shape[0]
    need = need or [(True, True)] * len(pairs)
    outs = [(torch.empty_like(q) if nq else None, torch.empty_like(k) if nk else None) for (q, k), (nq, nk) in zip(pairs, need)]
    dscale = torch.zeros(1, device=Q0.device, dtype=torch.float32)
    arr = _pair_array([(q, k, dq, dk) for (q, k), (dq, dk) in zip(pairs, outs)])
    check(L.load().dh_infonce_bwd(arr, len(pairs), b, B, D, ptr(scale), int(label0), _int_array(label0s, len(pairs)),
                                  _int_array(excl0s, len(pairs)), ptr(_contig(row_lse, "lse")), ptr(_contig(g_row, "g")),
                                  ptr(dscale), stream()), "dh_infonce_bwd")
    return outs, dscale


def ce_fused_ok(rows, weight):
    """can dh_ce_fused_fwd / bwd take the head?  (bf16 features [*, K] with K a multiple of 64; the caller pads the rows to 256)"""
    import os
    if os.environ.get("DH_CE_FUSED", "1") == "0":
        return False
    return (rows.is_cuda and rows.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16
            and rows.shape[1] % 64 == 0 and rows.shape[1] >= 128 and weight.shape[0] >= 256)


def ce_fused_fwd(rows, weight, bias, labels, n):
    """rows [n_pad, K] bf16 (rows >= n zero), weight [V, K] bf16, bias [V] fp32, labels [n] int64 -> row_loss, row_lse [n] fp32."""
    n_pad, K = rows.shape
    V = weight.shape[0]
    _req(rows.is_contiguous() and weight.stride(1) == 1 and weight.stride(0) == K and labels.dtype == torch.int64 and labels.numel() == n, "ce_fused_fwd operands")
    row_loss = torch.empty(n, device=rows.device, dtype=torch.float32)
    row_lse = torch.empty(n, device=rows.device, dtype=torch.float32)
    lib = L.load()
    nbytes = lib.dh_ce_fused_ws_bytes(n_pad, V)
    ws = torch.empty(nbytes // 4, device=rows.device, dtype=torch.float32)
    check(lib.dh_ce_fused_fwd(ptr(rows), ptr(weight), ptr(bias), ptr(labels), n, n_pad, V, K, ptr(row_loss), ptr(row_lse), ptr(ws), nbytes,
                              stream()), "dh_ce_fused_fwd")
    return row_loss, row_lse


def ce_fused_bwd(rows, weight, bias, labels, row_lse, g_row, n, ldd):
    """dl [n_pad, ldd] bf16 = g_row * (softmax(rows W^T + bias) - onehot(labels)); zero rows >= n / columns >= V."""
    n_pad, K = rows.shape
    V = weight.shape[0]
    dl = torch.empty(n_pad, ldd, device=rows.device, dtype=torch.bfloat16)
    if ldd > (V + 255) // 256 * 256:
        dl.zero_()
    check(L.load().dh_ce_fused_bwd(ptr(rows), ptr(weight), ptr(bias), ptr(labels), ptr(_contig(row_lse, "lse")), ptr(_contig(g_row, "g")), n, n_pad,
                                   V, K, ptr(dl), ldd, stream()), "dh_ce_fused_bwd")
    return dl


def ce_rows_fwd(logits, labels):
    _req(logits.dtype == torch.float32 and logits.dim() == 2 and logits.stride(1) == 1, 'logits.dtype == torch.float32 and logits.dim() == 2 and logits.stride(1) == 1')
    rows, C = logits.shape
    mk = lambda: torch.empty(rows, device=logits.device, dtype=torch.float32)
    row_loss, row_lse, c1, c5 = mk(), mk(), mk(), mk()
    check(L.load().dh_ce_rows_fwd(ptr(logits), logits.stride(0), ptr(labels), rows, C, ptr(row_loss), ptr(row_lse), ptr(c1),
                                  ptr(c5), stream()), "dh_ce_rows_fwd")
    return row_loss, row_lse, c1, c5


def ce_rows_bwd(logits, labels, row_lse, g_row):
    rows, C = logits.shape
    dlogits = torch.empty(rows, C, device=logits.device, dtype=torch.float32)
    check(L.load().dh_ce_rows_bwd(ptr(logits), logits.stride(0), ptr(labels), rows, C, ptr(row_lse), ptr(_contig(g_row, "g")),
                                  ptr(dlogits), C, stream()), "dh_ce_rows_bwd")
    return dlogits


def adamw(p, g, m, v, p_bf16, lr, beta1, beta2, eps, wd, step, grad_scale=1.0):
    n = p.numel()
    check(L.load().dh_adamw(ptr(p), ptr(g), ptr(m), ptr(v), ptr(p_bf16), n, lr, beta1, beta2, eps, wd, step, grad_scale,
                            stream()), "dh_adamw")


def adamw_segmented(p, g, m, v, p_bf16, seg_start, seg_lr, seg_wd, beta1, beta2, eps, step, grad_scale=1.0):
    """One launch over the whole flat buffer; per-segment (lr, wd) tables live on the device."""
    n = p.numel()
    check(L.load().dh_adamw_segmented(ptr(p), ptr(g), ptr(m), ptr(v), ptr(p_bf16), n, ptr(seg_start), ptr(seg_lr),
                                      ptr(seg_wd), seg_start.numel(), beta1, beta2, eps, step, grad_scale, stream()),
          "dh_adamw_segmented")


def cast(src, dst):
    _req(src.numel() == dst.numel(), 'src.numel() == dst.numel()')
    check(L.load().dh_cast(dt(src), ptr(src), dt(dst), ptr(dst), src.numel(), stream()), "dh_cast")
    return dst


def ce_rows_bwd_padded(logits, labels, row_lse, g_row, C, out_dtype, rows_pad, C_pad):
    """dlogits [rows_pad, C_pad] (out_dtype), zero outside [rows) x [C) -- padded layout for the MLM GEMMs."""
    rows = labels.numel()
    dl = torch.empty(rows_pad, C_pad, device=logits.device, dtype=out_dtype)
    check(L.load().dh_ce_rows_bwd_padded(ptr(logits), logits.stride(0), ptr(labels), rows, C, ptr(row_lse), ptr(_contig(g_row, "g")),
                                         ptr(dl), dt(dl), C_pad, rows_pad, C_pad, stream()), "dh_ce_rows_bwd_padded")
    return dl


def bn1d_fwd(x, w, b, running_mean, running_var, groups, relu, training, eps=1e-5, momentum=0.1):
    _contig(x, "x")
    rows, C = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(groups, C, device=x.device, dtype=torch.float32)
    invstd = torch.empty(groups, C, device=x.device, dtype=torch.float32)
    check(L.load().dh_bn1d_fwd(dt(x), ptr(x), ptr(w), ptr(b), ptr(y), ptr(mean), ptr(invstd), ptr(running_mean), ptr(running_var),
                               groups, rows // groups, C, eps, momentum, int(relu), int(training), stream()), "dh_bn1d_fwd")
    return y, mean, invstd


def bn1d_bwd(dy, x, y, w, mean, invstd, dw, db, groups, relu):
    _contig(dy, "dy")
    rows, C = x.shape
    dx = torch.empty_like(x)
    check(L.load().dh_bn1d_bwd(dt(x), ptr(dy), ptr(x), ptr(y), ptr(w), ptr(mean), ptr(invstd), ptr(dx), ptr(dw), ptr(db), groups,
                               rows // groups, C, int(relu), stream()), "dh_bn1d_bwd")
    return dx


def cos_rows_fwd(p, z):
    _contig(p, "p"), _contig(z, "z")
    rows, d = p.shape
    out = torch.empty(rows, device=p.device, dtype=torch.float32)
    check(L.load().dh_cos_rows_fwd(dt(p), ptr(p), ptr(z), ptr(out), rows, d, stream()), "dh_cos_rows_fwd")
    return out


def cos_rows_bwd(p, z, g_row):
    rows, d = p.shape
    dp = torch.empty_like(p)
    check(L.load().dh_cos_rows_bwd(dt(p), ptr(p), ptr(z), ptr(_contig(g_row, "g")), ptr(dp), rows, d, stream()), "dh_cos_rows_bwd")
    return dp


def nn_bank_query(q, bank):
    """q [rows, D] fp32, bank [size, D] fp32 -> (idx [rows] int64, feats [rows, D] = bank[idx])."""
    _contig(q, "q"), _contig(bank, "bank")
    rows, D = q.shape
    size = bank.shape[0]
    lib = L.load()
    nbytes = lib.dh_nn_bank_ws_bytes(rows, size)
    ws = torch.empty(nbytes // 4, device=q.device, dtype=torch.float32)
    idx = torch.empty(rows, device=q.device, dtype=torch.int64)
    feats = torch.empty(rows, D, device=q.device, dtype=torch.float32)
    check(lib.dh_nn_bank_query(ptr(q), ptr(bank), rows, size, D, ptr(idx), ptr(feats), ptr(ws), nbytes, stream()), "dh_nn_bank_query")
    return idx, feats


def nn_bank_enqueue(store, ptr_dev, batch, size):
    """FIFO enqueue of the NN queue behind the C-ABI (dh_nn_bank_enqueue): store [size + spill, D] fp32, ptr_dev int64 [1] on the device."""
    _req(store.dtype == torch.float32 and batch.dtype == torch.float32 and store.is_contiguous() and batch.is_contiguous() and
         ptr_dev.dtype == torch.int64 and ptr_dev.numel() == 1 and store.shape[1] == batch.shape[1], "nn_bank_enqueue: fp32 [rows, D], int64 [1]")
    check(L.load().dh_nn_bank_enqueue(ptr(store), ptr(ptr_dev), ptr(batch), batch.shape[0], int(size), store.shape[0] - int(size), store.shape[1],
                                      stream()), "dh_nn_bank_enqueue")


def gather_rows(x, idx, n_pad=None):
    n = idx.numel()
    n_pad = n_pad or max(n, 1)
    d = x.shape[-1]
    out = torch.empty(n_pad, d, device=x.device, dtype=x.dtype)
    check(L.load().dh_gather_rows(dt(x), ptr(x), ptr(idx), ptr(out), n, n_pad, d, stream()), "dh_gather_rows")
    return out


def scatter_rows_add(dout, idx, dx):
    n = idx.numel()
    check(L.load().dh_scatter_rows_add(dt(dx), ptr(dout), ptr(idx), ptr(dx), n, dx.shape[-1], stream()), "dh_scatter_rows_add")
    return dx


def filip_select(img_tok, txt_tok):
    """img_tok [b,J,D], txt_tok [b,T,D] fp32 normalised -> (idx_img [b,16], idx_txt [b,16]) int64."""
    _contig(img_tok, "img_tok"), _contig(txt_tok, "txt_tok")
    b, J, D = img_tok.shape
    T = txt_tok.shape[1]
    ia = torch.empty(b, 16, device=img_tok.device, dtype=torch.int64)
    ib = torch.empty(b, 16, device=img_tok.device, dtype=torch.int64)
    check(L.load().dh_filip_select(ptr(img_tok), ptr(txt_tok), b, J, T, D, ptr(ia), ptr(ib), stream()), "dh_filip_select")
    return ia, ib


def maxsim_reduce(S, b, B, J, scale):
    """S [b*J, B*16] -> logits [b,B] (scaled), raw [b,B], argmax [b*J, B] uint8."""
    _req(S.dim() == 2 and S.stride(1) == 1 and S.shape[0] == b * J and S.shape[1] >= B * 16, 'S.dim() == 2 and S.stride(1) == 1 and S.shape[0] == b * J and S.shape[1] >= B * 16')
    logits = torch.empty(b, B, device=S.device, dtype=torch.float32)
    raw = torch.empty(b, B, device=S.device, dtype=torch.float32)
    arg = torch.empty(b * J, B, device=S.device, dtype=torch.uint8)
    check(L.load().dh_maxsim_reduce(dt(S), ptr(S), S.stride(0), b, B, J, ptr(scale), ptr(logits), ptr(raw), ptr(arg), stream()),
          "dh_maxsim_reduce")
    return logits, raw, arg


def maxsim_fused_ok(Q, K, B, J):
    """can dh_maxsim_fused_fwd take this problem?  (bf16 token features, whole 256-tiles after row padding)"""
    import os
    D = Q.shape[1]
    if os.environ.get("DH_MAXSIM_FUSED", "1") == "0":          # A/B switch: scores in caption chunks + dh_maxsim_reduce instead
        return False
    return Q.is_cuda and Q.dtype == torch.bfloat16 and K.dtype == torch.bfloat16 and B % 16 == 0 and D % 64 == 0 and D >= 128 and J >= 19


def maxsim_fused_fwd(Q, K, b, B, J, scale):
    """Q [rows_pad, D] bf16 (rows_pad % 256 == 0, rows >= b*J), K [B*16, D] bf16 -> logits [b,B], raw [b,B], argmax [rows_pad, B] uint8."""
    _req(Q.dim() == 2 and K.dim() == 2 and Q.is_contiguous() and K.is_contiguous() and K.shape == (B * 16, Q.shape[1]), 'Q.dim() == 2 and K.dim() == 2 and Q.is_contiguous() and K.is_contiguous() and K.shape == (B * 16, Q.shape[1])')
    rows_pad = Q.shape[0]
    logits = torch.empty(b, B, device=Q.device, dtype=torch.float32)
    raw = torch.empty(b, B, device=Q.device, dtype=torch.float32)
    arg = torch.empty(rows_pad, B, device=Q.device, dtype=torch.uint8)
    check(L.load().dh_maxsim_fused_fwd(ptr(Q), ptr(K), rows_pad, b, B, J, Q.shape[1], ptr(scale), ptr(logits), ptr(raw), ptr(arg), stream()),
          "dh_maxsim_fused_fwd")
    return logits, raw, arg


def maxsim_scatter_rows(dlogits, arg, scale, b, B, J, r0, nrows, out):
    """rows [r0, r0 + nrows) of the one-hot-weighted G into the chunk buffer `out` [>= nrows, B*16]."""
    _req(out.dim() == 2 and out.stride(1) == 1 and out.shape[0] >= nrows and out.shape[1] >= B * 16 and arg.shape[0] >= min(r0 + nrows, b * J), 'out.dim() == 2 and out.stride(1) == 1 and out.shape[0] >= nrows and out.shape[1] >= B * 16 and arg.shape[0] >= min(r0 + nrows, b * J)')
    check(L.load().dh_maxsim_scatter_rows(dt(out), ptr(_contig(dlogits, "dlogits")), ptr(arg), ptr(scale), ptr(out), out.stride(0), b, B, J,
                                          r0, nrows, stream()), "dh_maxsim_scatter_rows")
    return out[:nrows]


def maxsim_scatter(dlogits, arg, scale, b, B, J, dtype):
    G = torch.empty(b * J, B * 16, device=dlogits.device, dtype=dtype)
    check(L.load().dh_maxsim_scatter(dt(G), ptr(_contig(dlogits, "dlogits")), ptr(arg), ptr(scale), ptr(G), G.stride(0), b, B, J,
                                     stream()), "dh_maxsim_scatter")
    return G


# ---------------------------------------------------------------------------------------------
# ModifiedResNet tower (csrc/resnet_ops.hip): NHWC activations as [N*H*W, C] pixel rows
# ---------------------------------------------------------------------------------------------
def conv_rows(x, N, H, W, C, stride=1, pad=1, out=None):
    """3x3 patches of an NHWC activation x [N*H*W, C] -> rows [N*Ho*Wo, 9*C], inner order (c, ky, kx)
    (== conv.weight.view(Cout, Cin*9))."""
    _contig(x, "x")
    _req(x.shape == (N * H * W, C), 'x.shape == (N * H * W, C)')
    Ho, Wo = (H + 2 * pad - 3) // stride + 1, (W + 2 * pad - 3) // stride + 1
    if out is None:
        out = torch.empty(N * Ho * Wo, 9 * C, device=x.device, dtype=x.dtype)
    _req(out.shape == (N * Ho * Wo, 9 * C) and out.is_contiguous() and out.dtype == x.dtype, 'out.shape == (N * Ho * Wo, 9 * C) and out.is_contiguous() and out.dtype == x.dtype')
    check(L.load().dh_conv_rows(dt(x), ptr(x), 0, C, 0, ptr(out), N, H, W, C, 3, stride, pad, 9 * C, stream()), "dh_conv_rows")
    return out, Ho, Wo


def conv_rows_image(images, c0, dtype, stride=2, pad=1, out=None):
    """3x3 patches of the fp32 NCHW image batch (3 channels from c0) -> rows [N*Ho*Wo, 32] (K = 27 zero-padded)."""
    _contig(images, "images")
    _req(images.dtype == torch.float32 and images.dim() == 4, 'images.dtype == torch.float32 and images.dim() == 4')
    N, ct, H, W = images.shape
    Ho, Wo = (H + 2 * pad - 3) // stride + 1, (W + 2 * pad - 3) // stride + 1
    if out is None:
        out = torch.empty(N * Ho * Wo, 32, device=images.device, dtype=dtype)
    _req(out.shape == (N * Ho * Wo, 32) and out.is_contiguous(), 'out.shape == (N * Ho * Wo, 32) and out.is_contiguous()')
    check(L.load().dh_conv_rows(dt(out), ptr(images), 1, ct, c0, ptr(out), N, H, W, 3, 3, stride, pad, 32, stream()), "dh_conv_rows")
    return out, Ho, Wo


_BN_WS = {}


def _bn_ws(device, nbytes):
    key = (device.type, device.index, stream())
    ws = _BN_WS.get(key)
    if ws is None or ws.numel() * 4 < nbytes:
        ws = torch.empty((nbytes + 3) // 4, device=device, dtype=torch.float32)
        _BN_WS[key] = ws
    return ws


def bn2d_fwd(x, w, b, running_mean, running_var, relu, training, residual=None, eps=1e-5, momentum=0.1):
    """y = relu?(BatchNorm(x) (+ residual)) on pixel rows x [R, C]; returns (y, save_mean [C], save_invstd [C])."""
    _contig(x, "x")
    R, C = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(C, device=x.device, dtype=torch.float32)
    invstd = torch.empty(C, device=x.device, dtype=torch.float32)
    if residual is not None:
        _req(residual.shape == x.shape and residual.dtype == x.dtype and residual.is_contiguous(), 'residual.shape == x.shape and residual.dtype == x.dtype and residual.is_contiguous()')
    lib = L.load()
    nbytes = lib.dh_bn2d_ws_bytes(R, C)
    ws = _bn_ws(x.device, nbytes)
    check(lib.dh_bn2d_fwd(dt(x), ptr(x), ptr(residual), ptr(w), ptr(b), ptr(y), ptr(mean), ptr(invstd), ptr(running_mean),
                          ptr(running_var), R, C, eps, momentum, int(relu), int(training), ptr(ws), ws.numel() * 4, stream()),
          "dh_bn2d_fwd")
    return y, mean, invstd


def bn2d_bwd(dy, x, y, w, mean, invstd, dw, db, relu, want_dres=False):
    """-> dx (, dres = dy masked by the ReLU: the gradient of the residual branch); dw, db accumulate."""
    _contig(dy, "dy"), _contig(x, "x")
    R, C = x.shape
    _req(dy.shape == x.shape and dy.dtype == x.dtype and dw.dtype == torch.float32 and db.dtype == torch.float32, 'dy.shape == x.shape and dy.dtype == x.dtype and dw.dtype == torch.float32 and db.dtype == torch.float32')
    dx = torch.empty_like(x)
    dres = torch.empty_like(x) if want_dres else None
    lib = L.load()
    nbytes = lib.dh_bn2d_ws_bytes(R, C)
    ws = _bn_ws(x.device, nbytes)
    check(lib.dh_bn2d_bwd(dt(x), ptr(dy), ptr(x), ptr(y) if relu else None, ptr(w), ptr(mean), ptr(invstd), ptr(dx), ptr(dres),
                          ptr(dw), ptr(db), R, C, int(relu), ptr(ws), ws.numel() * 4, stream()), "dh_bn2d_bwd")
    return (dx, dres) if want_dres else dx


def bn2d_sums(x, dy=None, y=None, mean=None, invstd=None, relu=False):
    """[2C + 1] float64: per-channel (sum x, sum x^2) -- or, with dy, (sum dyr, sum dyr*xhat) -- and the row count."""
    _contig(x, "x")
    R, C = x.shape
    sums = torch.empty(2 * C + 1, device=x.device, dtype=torch.float64)
    lib = L.load()
    ws = _bn_ws(x.device, lib.dh_bn2d_ws_bytes(R, C))
    mode = 0 if dy is None else 1
    check(lib.dh_bn2d_sums(dt(x), mode, ptr(x), ptr(dy), ptr(y) if relu else None, ptr(mean), ptr(invstd), int(relu), R, C, ptr(sums),
                           ptr(ws), ws.numel() * 4, stream()), "dh_bn2d_sums")
    return sums


def bn2d_fwd_apply(x, w, b, sums, running_mean, running_var, relu, residual=None, eps=1e-5, momentum=0.1):
    _contig(x, "x")
    R, C = x.shape
    _req(sums.dtype == torch.float64 and sums.numel() == 2 * C + 1, 'sums.dtype == torch.float64 and sums.numel() == 2 * C + 1')
    y = torch.empty_like(x)
    mean = torch.empty(C, device=x.device, dtype=torch.float32)
    invstd = torch.empty(C, device=x.device, dtype=torch.float32)
    check(L.load().dh_bn2d_fwd_apply(dt(x), ptr(x), ptr(residual), ptr(w), ptr(b), ptr(sums), ptr(y), ptr(mean), ptr(invstd),
                                     ptr(running_mean), ptr(running_var), R, C, eps, momentum, int(relu), stream()), "dh_bn2d_fwd_apply")
    return y, mean, invstd


def bn2d_bwd_apply(dy, x, y, w, mean, invstd, sums_local, sums_global, dw, db, relu, want_dres=False):
    _contig(dy, "dy"), _contig(x, "x")
    R, C = x.shape
    dx = torch.empty_like(x)
    dres = torch.empty_like(x) if want_dres else None
    ws = _bn_ws(x.device, 8 * C)
    check(L.load().dh_bn2d_bwd_apply(dt(x), ptr(dy), ptr(x), ptr(y) if relu else None, ptr(w), ptr(mean), ptr(invstd), ptr(sums_local),
                                     ptr(sums_global), ptr(dx), ptr(dres), ptr(dw), ptr(db), R, C, int(relu), ptr(ws), ws.numel() * 4,
                                     stream()), "dh_bn2d_bwd_apply")
    return (dx, dres) if want_dres else dx


def avgpool_fwd(x, N, H, W, C, k):
    _contig(x, "x")
    _req(x.shape == (N * H * W, C), 'x.shape == (N * H * W, C)')
    y = torch.empty(N * (H // k) * (W // k), C, device=x.device, dtype=x.dtype)
    check(L.load().dh_avgpool_fwd(dt(x), ptr(x), ptr(y), N, H, W, C, k, stream()), "dh_avgpool_fwd")
    return y


def avgpool_bwd(dy, N, H, W, C, k):
    _contig(dy, "dy")
    _req(dy.shape == (N * (H // k) * (W // k), C), 'dy.shape == (N * (H // k) * (W // k), C)')
    dx = torch.empty(N * H * W, C, device=dy.device, dtype=dy.dtype)
    check(L.load().dh_avgpool_bwd(dt(dy), ptr(dy), ptr(dx), N, H, W, C, k, stream()), "dh_avgpool_bwd")
    return dx


def attnpool_tokens_fwd(x, pos, b, HW):
    _contig(x, "x")
    C = x.shape[1]
    _req(x.shape[0] == b * HW and pos.shape == (HW + 1, C) and pos.dtype == torch.float32 and pos.is_contiguous(), 'x.shape[0] == b * HW and pos.shape == (HW + 1, C) and pos.dtype == torch.float32 and pos.is_contiguous()')
    tok = torch.empty(b * (HW + 1), C, device=x.device, dtype=x.dtype)
    check(L.load().dh_attnpool_tokens_fwd(dt(x), ptr(x), ptr(pos), ptr(tok), b, HW, C, stream()), "dh_attnpool_tokens_fwd")
    return tok


def attnpool_tokens_bwd(dtok, dpos, b, HW):
    _contig(dtok, "dtok")
    C = dtok.shape[1]
    _req(dtok.shape[0] == b * (HW + 1) and (dpos is None or (dpos.shape == (HW + 1, C) and dpos.dtype == torch.float32)), 'dtok.shape[0] == b * (HW + 1) and (dpos is None or (dpos.shape == (HW + 1, C) and dpos.dtype == torch.float32))')
    dx = torch.empty(b * HW, C, device=dtok.device, dtype=dtok.dtype)
    check(L.load().dh_attnpool_tokens_bwd(dt(dtok), ptr(dtok), ptr(dx), ptr(dpos), b, HW, C, stream()), "dh_attnpool_tokens_bwd")
    return dx

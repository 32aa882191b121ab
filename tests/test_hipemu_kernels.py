"""The `-m gpu` kernel tests of tests/test_gpu_kernels.py, executed on the CPU: the SAME test functions (torch references,
tolerances) against the SAME C entry points and kernel sources, compiled for the host with HIP threads as fibers and the cross-lane
operations (shuffles, MFMA builtins, the LDS transpose read) as wave collectives (tests/hipemu).  Two things are checked at once:
the emulation's operand / accumulator layouts (these kernels pass on the hardware, so they must pass here), and -- from now on --
any edit of a kernel written in plain HIP C++ before a GPU is available.  Not covered: the GEMM families written with inline ISA
(gemm_v4 / gemm_v3 / gemm_glds decline on the host build; the MFMA-builtin tiles and the generic kernel of gemm.hip run instead)."""
import pytest
import torch

from hipemu_util import emulated_gpu

F32, BF16 = torch.float32, torch.bfloat16
CASES = [
    ("test_gemm_layouts", (BF16, False, False, False, 200, 136, 192)),
    ("test_gemm_layouts", (BF16, False, True, True, 128, 128, 64)),
    ("test_gemm_layouts", (BF16, False, False, True, 37, 264, 520)),
    ("test_gemm_layouts", (BF16, True, True, False, 37, 264, 520)),
    ("test_gemm_layouts", (F32, False, True, False, 200, 136, 192)),
    ("test_gemm_epilogues", (BF16,)),
    ("test_gemm_epilogues", (F32,)),
    ("test_gemm_weight_grad_accumulate_splitk", (BF16,)),
    ("test_gemm_weight_grad_accumulate_splitk", (F32,)),
    ("test_colsum", ()),
    ("test_layernorm", (BF16, 50, 768)),
    ("test_layernorm", (BF16, 51, 768)),
    ("test_layernorm", (F32, 7, 256)),
    ("test_layernorm", (F32, 77, 512)),
    ("test_layernorm", (BF16, 9, 100)),
    ("test_layernorm_deferred_reduce_of_many", (BF16,)),
    ("test_layernorm_deferred_reduce_of_many", (F32,)),
    ("test_attention", (BF16, 3, 50, 12, False)),
    ("test_attention", (BF16, 2, 77, 8, True)),
    ("test_attention", (BF16, 2, 33, 1, True)),
    ("test_attention", (F32, 2, 5, 2, False)),
    ("test_attention", (F32, 1, 16, 2, True)),
    ("test_text_embed", (BF16,)),
    ("test_vision_embed", (F32,)),
    ("test_embed_table_grad_sorted_segments", (700, 520, 90, F32)),
    ("test_embed_table_grad_sorted_segments", (1500, 512, 3000, BF16)),
    ("test_embed_table_grad_sorted_segments", (300, 64, 70001, F32)),
    ("test_pool_and_l2norm", (BF16,)),
    ("test_infonce", (8, 8, 64, 0)),
    ("test_infonce", (40, 120, 512, 40)),
    ("test_infonce", (33, 99, 768, 66)),
    ("test_infonce", (40, 140, 1280, 100)),
    ("test_ce_rows", ()),
    ("test_adamw_matches_torch", ()),
    ("test_bn1d_groups", (BF16, True, 2, 24, 200)),
    ("test_bn1d_groups", (F32, False, 2, 24, 203)),
    ("test_bn1d_groups", (F32, True, 3, 70, 72)),
    ("test_bn1d_groups", (BF16, False, 2, 128, 256)),
    ("test_cos_rows", (F32,)),
    ("test_nn_bank_query_ties_and_ragged_sizes", ()),
    ("test_nn_bank_query_exact", (40, 5000, 512)),
    ("test_gather_scatter_rows", (BF16,)),
    ("test_ce_rows_bwd_padded_layout", ()),
    ("test_filip_select_and_maxsim", ()),
]


@pytest.mark.parametrize("name,args", CASES, ids=["%s-%d" % (c[0], i) for i, c in enumerate(CASES)])
def test_gpu_kernel_test_on_host_emulation(monkeypatch, name, args):
    import test_gpu_kernels as T
    monkeypatch.setattr(T, "cuda", torch.device("cpu"))
    monkeypatch.setattr(T, "_poison_lds", lambda ops: None)      # the emulation NaN-poisons a block's dynamic LDS itself
    with emulated_gpu():
        getattr(T, name)(*args)


def _pil_resized_crop(img_u8, box, Wf, Hf, ox, oy, W, H, flip):
    """what the reference's workers do per image: PIL crop + resize(BILINEAR) (+ centre window, mirror), still uint8"""
    from PIL import Image
    x0, y0, w, h = box
    im = Image.fromarray(img_u8.numpy()).crop((x0, y0, x0 + w, y0 + h)).resize((Wf, Hf), Image.BILINEAR)
    im = im.crop((ox, oy, ox + W, oy + H))
    if flip:
        im = im.transpose(Image.FLIP_LEFT_RIGHT)
    import numpy as np
    return torch.from_numpy(np.asarray(im).copy())


def test_image_resized_crop_matches_pil_and_torch_antialias():
    """dh_image_resized_crop_u8 on the host emulation against (a) PIL itself -- the library the reference's torchvision transforms
    call -- within one grey level (PIL resamples in two uint8 passes with fixed-point weights), unbiased;
    (b) torch's antialiased bilinear on the crop box, which is the same filter in float, to 1e-4."""
    import numpy as np
    import torch.nn.functional as F
    from declip_amd import augment
    g = torch.Generator().manual_seed(0)
    sizes = [(300, 400), (250, 230), (512, 340), (224, 224), (97, 130)]
    Hs, Ws = 512, 400
    canvas = torch.zeros(len(sizes), Hs, Ws, 3, dtype=torch.uint8)
    for n, (h, w) in enumerate(sizes):
        # smooth structure + noise: pure noise would hide filter-phase errors behind +-1 rounding
        yy, xx = torch.meshgrid(torch.arange(h).float(), torch.arange(w).float(), indexing="ij")
        base = torch.stack([127 + 100 * torch.sin(xx / 9 + n), 127 + 100 * torch.cos(yy / 7), 127 + 60 * torch.sin((xx + yy) / 13)], dim=-1)
        canvas[n, :h, :w] = (base + 20 * torch.randn(h, w, 3, generator=g)).clamp(0, 255).to(torch.uint8)
    rng = np.random.default_rng(1)
    mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    cases = [("rrc", augment.random_resized_crop_params(sizes, (224, 224), generator=rng), (224, 224)),
             ("minsize", augment.random_crop_min_size_params(sizes, 224, generator=rng), (224, 224)),
             ("onecrop", augment.resize_center_crop_params(sizes[:4], 256, 224), (224, 224)),
             ("upscale", np.array([[10, 20, 50, 40, 96, 64, 0, 0]] * 2, dtype=np.int32), (64, 96))]
    for name, params, (H, W) in cases:
        b = params.shape[0]
        flip = torch.tensor([i % 2 for i in range(b)], dtype=torch.uint8)
        with emulated_gpu() as ops:
            out = ops.image_resized_crop_u8(canvas[:b].contiguous(), torch.from_numpy(params), (H, W), flip=flip, mean=mean, std=std)
            raw = ops.image_resized_crop_u8(canvas[:b].contiguous(), torch.from_numpy(params), (H, W), flip=flip, mean=(0, 0, 0), std=(1, 1, 1),
                                            round_u8=False)
        m, s = torch.tensor(mean).view(3, 1, 1), torch.tensor(std).view(3, 1, 1)
        for i in range(b):
            x0, y0, w, h, Wf, Hf, ox, oy = [int(v) for v in params[i]]
            levels = ((out[i] * s + m) * 255.0).round()                                    # back to grey levels
            pil = _pil_resized_crop(canvas[i], (x0, y0, w, h), Wf, Hf, ox, oy, W, H, bool(flip[i])).permute(2, 0, 1).float()
            # PIL rounds to uint8 after its horizontal pass and again after the vertical one: a quarter of the pixels land one level
            # away from the single-pass result, never two, and (but for exact .5 ties, which its fixed-point weights break downwards) without a sign
            d = levels - pil
            assert float(d.abs().max()) <= 1.0 and float(d.abs().mean()) <= 0.3 and abs(float(d.mean())) <= 0.1, \
                (name, i, float(d.abs().max()), float(d.abs().mean()), float(d.mean()))
            crop = canvas[i, y0:y0 + h, x0:x0 + w].permute(2, 0, 1).float()[None]
            ref = F.interpolate(crop, size=(Hf, Wf), mode="bilinear", antialias=True, align_corners=False)[0, :, oy:oy + H, ox:ox + W]
            if flip[i]:
                ref = ref.flip(2)
            assert float((raw[i] * 255.0 - ref).abs().max()) <= 2e-3, (name, i)


def test_make_canvas_and_full_intake_path():
    """decoded images of different sizes -> canvas -> boxes -> on-device crops == the oracle's per-image pipeline"""
    import numpy as np
    from declip_amd import augment
    from declip_amd.prefetch import crops_on_device
    from oracle import restated
    g = torch.Generator().manual_seed(9)
    imgs = [torch.randint(0, 256, (h, w, 3), generator=g, dtype=torch.uint8) for h, w in ((90, 120), (150, 80), (64, 64))]
    canvas, sizes = augment.make_canvas([imgs[0].numpy(), imgs[1], imgs[2]])
    assert canvas.shape == (3, 150, 120, 3) and sizes == [(90, 120), (150, 80), (64, 64)]
    assert torch.equal(canvas[1, :150, :80], imgs[1]) and int(canvas[0, 90:].max()) == 0
    params = augment.resize_center_crop_params(sizes, 72, 64)
    with emulated_gpu():
        out = crops_on_device({"images": canvas, "image_boxes": torch.from_numpy(params)}, (64, 64))["images"]
    ref = restated.image_resized_crop_u8(canvas, params, (64, 64))
    d = (out - ref).abs()
    assert float(d.max()) <= 1.01 / 255 / 0.224 and float((d > 1e-6).float().mean()) <= 2e-3
    with pytest.raises(ValueError):
        augment.make_canvas([torch.zeros(4, 4, dtype=torch.uint8)])


def test_crop_box_generators():
    import numpy as np
    from declip_amd import augment
    rng = np.random.default_rng(3)
    sizes = [(480, 640)] * 400 + [(50, 600)] * 10
    p = augment.random_resized_crop_params(sizes, (224, 224), scale=(0.5, 1.0), generator=rng)
    x0, y0, w, h = p[:, 0], p[:, 1], p[:, 2], p[:, 3]
    hs, ws = np.array([s[0] for s in sizes]), np.array([s[1] for s in sizes])
    assert (x0 >= 0).all() and (y0 >= 0).all() and (x0 + w <= ws).all() and (y0 + h <= hs).all()
    frac = (w[:400] * h[:400]) / (480.0 * 640.0)
    assert 0.49 <= frac.min() and frac.max() <= 1.0 and 0.60 <= frac.mean() <= 0.80            # area ~ U(0.5, 1), large non-square boxes are redrawn
    asp = w[:400] / h[:400]
    assert 0.74 <= asp.min() and asp.max() <= 1.34
    assert (p[400:, 2] == 67).all() and (p[400:, 3] == 50).all()          # 12:1 images: fallback box, aspect clamped to 4/3
    q = augment.random_crop_min_size_params([(300, 500), (500, 300), (64, 64)], 224, generator=rng)
    assert q[0, 2] == q[0, 3] == 300 and q[0, 1] == 0 and 0 <= q[0, 0] <= 200
    assert q[1, 2] == q[1, 3] == 300 and q[1, 0] == 0 and 0 <= q[1, 1] <= 200 and tuple(q[2, :4]) == (0, 0, 64, 64)
    r = augment.resize_center_crop_params([(480, 640), (640, 480)], 256, 224)
    assert tuple(r[0]) == (0, 0, 640, 480, 341, 256, 58, 16) and tuple(r[1]) == (0, 0, 480, 640, 256, 341, 16, 58)


def test_prefetcher_crops_views_from_one_upload():
    """prefetch.crops_on_device: two views (DeCLIP) cut from ONE uint8 canvas -> channel-stacked fp32 [b, 6, H, W]; view v equals a
    single-view call with that view's boxes / flips."""
    import numpy as np
    from declip_amd import augment
    from declip_amd.prefetch import crops_on_device
    g = torch.Generator().manual_seed(2)
    sizes = [(120, 160), (200, 140), (96, 96)]
    canvas = torch.randint(0, 256, (3, 200, 160, 3), generator=g, dtype=torch.uint8)
    rng = np.random.default_rng(5)
    boxes = np.stack([augment.random_resized_crop_params(sizes, (64, 64), generator=rng) for _ in range(2)], axis=1)     # [b, 2, 8]
    flips = torch.tensor([[0, 1], [1, 1], [0, 0]], dtype=torch.uint8)
    with emulated_gpu() as ops:
        out = crops_on_device({"images": canvas, "image_boxes": torch.from_numpy(boxes), "image_flip": flips, "captions": "kept"}, (64, 64))
        assert out["captions"] == "kept" and "image_boxes" not in out and out["images"].shape == (3, 6, 64, 64)
        for v in range(2):
            one = ops.image_resized_crop_u8(canvas, torch.from_numpy(np.ascontiguousarray(boxes[:, v])), (64, 64), flip=flips[:, v].contiguous())
            assert torch.equal(out["images"][:, 3 * v:3 * v + 3], one)
        assert crops_on_device({"images": canvas}, (64, 64))["images"] is canvas                  # nothing to do without boxes


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("causal", [True, False])
def test_attention_varlen_matches_per_sequence_reference(dtype, causal):
    """dh_attn_varlen_fwd / _bwd on packed rows (sequence i = rows cu[i] .. cu[i+1]) against dense attention computed sequence by
    sequence in fp32; lengths 1 .. Lmax incl. one at the maximum, one of a single token; rows outside the sequences untouched."""
    torch.manual_seed(0)
    heads, hd, Lmax = 3, 64, 77
    lens = [5, 77, 1, 33, 16, 64]
    b, d = len(lens), heads * hd
    rows = sum(lens)
    rows_pad = rows + 11
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32)
    qkv = (torch.randn(rows_pad, 3 * d) * 0.7).to(dtype)
    dout = torch.randn(rows_pad, d).to(dtype)
    with emulated_gpu() as ops:
        out, lse = ops.attn_varlen_fwd(qkv, cu, rows, b, Lmax, heads, causal)
        dqkv = ops.attn_varlen_bwd(qkv, out, dout, lse, cu, rows, b, Lmax, heads, causal)
    assert float(out[rows:].float().abs().max()) == 0.0 and float(dqkv[rows:].float().abs().max()) == 0.0
    tol = 2e-5 if dtype == F32 else 2e-2
    for i, n in enumerate(lens):
        r0 = int(cu[i])
        x = qkv[r0:r0 + n].float().requires_grad_()
        q, k, v = [t.reshape(n, heads, hd).transpose(0, 1) for t in x.split(d, dim=1)]
        s = (q @ k.transpose(1, 2)) * hd ** -0.5
        if causal:
            s = s + torch.full((n, n), float("-inf")).triu_(1)
        ref = (torch.softmax(s, dim=-1) @ v).transpose(0, 1).reshape(n, d)
        ref.backward(dout[r0:r0 + n].float())
        assert float((out[r0:r0 + n].float() - ref.detach()).abs().max()) <= tol * max(1.0, float(ref.abs().max())), (i, n)
        assert float((dqkv[r0:r0 + n].float() - x.grad).abs().max()) <= tol * max(1.0, float(x.grad.abs().max())) * (1 if dtype == F32 else 3), (i, n)
        ref_lse = torch.logsumexp(s.detach(), dim=-1)
        assert float((lse[i, :, :n] - ref_lse).abs().max()) <= (1e-4 if dtype == F32 else 3e-2)


@pytest.mark.parametrize("lens", [[5, 77, 1, 33, 48, 64, 49, 16, 20, 70], [60, 77, 50, 49], [3, 48, 17]])
def test_attention_length_buckets_and_device_side_rows_on_the_emulation(lens):
    """dh_attn_bucketed_fwd / _bwd (round 4): captions of at most 48 tokens on the 3-key-block instantiation, the rest on the 5-block one,
    the sequence lists, the two (start, count) ranges AND the valid row count (rows = -1 -> cu_seqlens[b]) read on the device -- against
    dh_attn_varlen_* with host-side rows on the same packed batch: same values, zero padding rows, lse equal on the valid entries; a mixed
    batch, one with an empty short bucket, one with an empty long bucket."""
    torch.manual_seed(4)
    heads, hd, Lmax, Ls = 2, 64, 77, 48
    b, d = len(lens), heads * hd
    rows = sum(lens)
    rows_pad = (rows + 255) // 256 * 256
    lt = torch.tensor(lens)
    cu = torch.tensor([0] + list(lt.cumsum(0)), dtype=torch.int32)
    short = lt <= Ls
    order = torch.sort((~short).to(torch.int32), stable=True)[1].to(torch.int32)
    ns = int(short.sum())
    ranges = torch.tensor([0, ns, ns, b - ns], dtype=torch.int32)
    qkv = (torch.randn(rows_pad, 3 * d) * 0.7).to(BF16)
    dout = torch.randn(rows_pad, d).to(BF16)
    with emulated_gpu() as ops:
        o0, l0 = ops.attn_varlen_fwd(qkv, cu, rows, b, Lmax, heads, True)
        g0 = ops.attn_varlen_bwd(qkv, o0, dout, l0, cu, rows, b, Lmax, heads, True)
        o1, l1 = ops.attn_bucketed_fwd(qkv, cu, order, ranges, -1, b, Lmax, Ls, heads, True)
        g1 = ops.attn_bucketed_bwd(qkv, o1, dout, l1, cu, order, ranges, -1, b, Lmax, Ls, heads, True)
        o2, _ = ops.attn_varlen_fwd(qkv, cu, -1, b, Lmax, heads, True)                      # plain kernels, device-side row count
    assert float(o1[rows:].float().abs().max()) == 0.0 and float(g1[rows:].float().abs().max()) == 0.0 and float(o2[rows:].float().abs().max()) == 0.0
    assert torch.equal(o2[:rows], o0[:rows])
    assert float((o1.float() - o0.float()).abs().max()) <= 1e-6 * float(o0.float().abs().max())
    assert float((g1.float() - g0.float()).abs().max()) <= 1e-6 * float(g0.float().abs().max())
    for i, n in enumerate(lens):
        assert torch.equal(l1[i, :, :n], l0[i, :, :n])


@pytest.mark.parametrize("L,causal", [(50, False), (64, True), (33, True)])
def test_attention_dense_mfma_matches_reference(L, causal):
    """dh_attn_fwd / _bwd, bf16 (the MFMA kernels; L = 50 / 64: four 16-key blocks, the image tower's instantiation; L = 33: three, with
    a half-empty last contraction step) against fp32 attention; six (batch, head) pairs on the emulated chip's workgroups."""
    torch.manual_seed(1)
    b, heads, hd = 3, 2, 64
    d = heads * hd
    qkv = (torch.randn(b * L, 3 * d) * 0.7).to(BF16)
    dout = torch.randn(b * L, d).to(BF16)
    with emulated_gpu() as ops:
        out, lse = ops.attn_fwd(qkv, b, L, heads, causal)
        dqkv = ops.attn_bwd(qkv, out, dout, lse, b, L, heads, causal)
    x = qkv.float().requires_grad_()
    q, k, v = [t.reshape(b, L, heads, hd).transpose(1, 2) for t in x.split(d, dim=1)]
    s = (q @ k.transpose(2, 3)) * hd ** -0.5
    if causal:
        s = s + torch.full((L, L), float("-inf")).triu_(1)
    ref = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(b * L, d)
    ref.backward(dout.float())
    assert float((out.float() - ref.detach()).abs().max()) <= 2e-2 * max(1.0, float(ref.abs().max()))
    assert float((dqkv.float() - x.grad).abs().max()) <= 6e-2 * max(1.0, float(x.grad.abs().max()))
    assert float((lse - torch.logsumexp(s.detach(), dim=-1)).abs().max()) <= 3e-2


@pytest.mark.parametrize("ordered", [False, True])
@pytest.mark.parametrize("dtype", [F32, BF16])
def test_attention_pooled_query_matches_reference(dtype, ordered):
    """dh_attn_pooled_fwd / _bwd: one query per sequence against its keys (kv rows row0 .. row0 + nkeys - 1): outputs, lse, dq and
    the dkv rows; rows that belong to no sequence (padding between sequences) stay zero -- ordered: zeroed by the launch itself
    (total_rows > 0, the buffer is handed over uninitialised), else by the caller's fill."""
    torch.manual_seed(1)
    heads, hd = 3, 64
    d = heads * hd
    nkeys = [1, 77, 50, 128, 9]
    gaps = [0, 3, 0, 27, 5]                                 # dense layouts leave unused rows between the sequences
    row0, r = [], 0
    for n, g in zip(nkeys, gaps):
        row0.append(r)
        r += n + g
    rows, b = r, len(nkeys)
    q = (torch.randn(b, d) * 0.8).to(dtype)
    kv = (torch.randn(rows, 2 * d) * 0.8).to(dtype)
    dout = torch.randn(b, d).to(dtype)
    r0_t, n_t = torch.tensor(row0, dtype=torch.int32), torch.tensor(nkeys, dtype=torch.int32)
    with emulated_gpu() as ops:
        out, lse = ops.attn_pooled_fwd(q, kv, r0_t, n_t, heads, 128)
        dq, dkv = ops.attn_pooled_bwd(q, kv, dout, lse, r0_t, n_t, heads, 128, ordered=ordered)
    tol = 2e-5 if dtype == F32 else 2e-2
    owned = torch.zeros(rows, dtype=torch.bool)
    for i, (r0, n) in enumerate(zip(row0, nkeys)):
        owned[r0:r0 + n] = True
        qi = q[i].float().requires_grad_()
        kvi = kv[r0:r0 + n].float().requires_grad_()
        k, v = kvi[:, :d].reshape(n, heads, hd), kvi[:, d:].reshape(n, heads, hd)
        s = torch.einsum("hc,nhc->hn", qi.reshape(heads, hd), k) * hd ** -0.5
        ref = torch.einsum("hn,nhc->hc", torch.softmax(s, dim=-1), v).reshape(d)
        ref.backward(dout[i].float())
        assert float((out[i].float() - ref.detach()).abs().max()) <= tol * max(1.0, float(ref.abs().max())), i
        assert float((lse[i] - torch.logsumexp(s.detach(), dim=-1)).abs().max()) <= (1e-4 if dtype == F32 else 3e-2)
        assert float((dq[i].float() - qi.grad).abs().max()) <= tol * 3 * max(1.0, float(qi.grad.abs().max())), i
        assert float((dkv[r0:r0 + n].float() - kvi.grad).abs().max()) <= tol * 3 * max(1.0, float(kvi.grad.abs().max())), i
    assert float(dkv[~owned].float().abs().max()) == 0.0

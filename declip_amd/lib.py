"""ctypes binding of libdeclip_hip.so (the C-ABI declared in include/declip_hip.h).

The product path has NO fallback: if the shared library is missing or a call
fails, a DeclipHipError is raised (never a silent PyTorch/CPU path).
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int64, c_void_p

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DECLIP_HIP_LIB") or os.path.join(HERE, "libdeclip_hip.so")    # override: ablation builds (tools/build_abl.sh)

DH_F32, DH_BF16 = 0, 1
EPI_NONE, EPI_GELU, EPI_DGELU = 0, 1, 2
MAX_PAIRS = 16


class DeclipHipError(RuntimeError):
    pass


class GemmArgs(Structure):
    _fields_ = [
        ("dtype", c_int), ("c_dtype", c_int), ("a_kmajor", c_int), ("b_kmajor", c_int),
        ("M", c_int), ("N", c_int), ("K", c_int),
        ("A", c_void_p), ("lda", c_int64), ("B", c_void_p), ("ldb", c_int64), ("C", c_void_p), ("ldc", c_int64),
        ("bias", c_void_p), ("epilogue", c_int), ("residual", c_void_p), ("ldr", c_int64),
        ("aux", c_void_p), ("ldaux", c_int64), ("accumulate", c_int), ("split_k", c_int), ("alpha", c_float),
        ("force_generic", c_int), ("a_colsum", c_void_p), ("pad_ok", c_int),
        ("ws", c_void_p), ("ws_bytes", c_int64),
    ]


class NcePair(Structure):
    _fields_ = [("Q", c_void_p), ("K", c_void_p), ("dQ", c_void_p), ("dK", c_void_p)]


class LnPart(Structure):
    """dh_ln_part (include/declip_hip.h): the partials of one LayerNorm backward waiting for dh_ln_reduce_many."""
    _fields_ = [("part", c_void_p), ("dw", c_void_p), ("db", c_void_p), ("nb", ctypes.c_int32), ("d", ctypes.c_int32)]


class BlockParams(Structure):
    """dh_block_params (include/declip_hip.h)"""
    _fields_ = ([(n, c_void_p) for n in ("w_in", "w_out", "w_fc", "w_proj", "b_in", "b_out", "b_fc", "b_proj", "ln1_w", "ln1_b", "ln2_w", "ln2_b",
                                          "g_w_in", "g_w_out", "g_w_fc", "g_w_proj", "g_b_in", "g_b_out", "g_b_fc", "g_b_proj",
                                          "g_ln1_w", "g_ln1_b", "g_ln2_w", "g_ln2_b")]
                + [("eps1", c_float), ("eps2", c_float)])


class BlockArgs(Structure):
    """dh_block_args (include/declip_hip.h)"""
    _fields_ = [("dtype", c_int), ("rows", c_int), ("d", c_int), ("heads", c_int), ("b", c_int), ("L", c_int), ("causal", c_int), ("save", c_int),
                ("cu", c_void_p), ("rows_valid", c_int), ("seq_order", c_void_p), ("seq_ranges", c_void_p), ("L_short", c_int), ("p", BlockParams), ("x", c_void_p), ("x_out", c_void_p),
                ("act", c_void_p), ("act_bytes", c_int64), ("ws", c_void_p), ("ws_bytes", c_int64), ("dx_out", c_void_p), ("dx", c_void_p),
                ("scratch", c_void_p), ("scratch_bytes", c_int64), ("ln_part1", c_void_p), ("ln_part2", c_void_p), ("ln_part_bytes", c_int64),
                ("ln_nb1", c_int), ("ln_nb2", c_int), ("dw_first_touch", c_int)]


_P = c_void_p
_PROTOS = {
    "dh_bpe_create": (c_void_p, [c_char_p, c_int64, c_int]),
    "dh_bpe_destroy": (None, [c_void_p]),
    "dh_bpe_vocab_size": (c_int, [c_void_p]),
    "dh_bpe_encode": (c_int, [c_void_p, c_char_p, _P, c_int, c_int, _P, _P, c_int]),
    "dh_last_error": (c_char_p, []),
    "dh_version": (c_int, []),
    "dh_device_info": (c_int, [c_int, POINTER(c_int)]),
    "dh_stream_abandon_capture": (c_int, [_P]),
    "dh_gemm": (c_int, [POINTER(GemmArgs), _P]),
    "dh_gemm_group": (c_int, [POINTER(GemmArgs), c_int, _P]),
    "dh_gemm_v4_enable": (c_int, [c_int]),
    "dh_gemm_v4_set_dynamic": (c_int, [c_int]),
    "dh_gemm_stats": (c_int, [_P, c_int]),
    "dh_colsum": (c_int, [c_int, _P, c_int64, c_int, c_int, _P, c_int, _P]),
    "dh_layernorm_fwd": (c_int, [c_int, _P, _P, _P, _P, _P, _P, c_int, c_int, c_float, _P]),
    "dh_zero_ranges": (c_int, [_P, _P, c_int, c_int64, _P]),
    "dh_layernorm_bwd_ws_bytes": (c_int64, [c_int, c_int]),
    "dh_layernorm_bwd": (c_int, [c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, _P, c_int64, _P]),
    "dh_layernorm_bwd_part": (c_int, [c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, _P, c_int64, _P, _P]),
    "dh_ln_reduce_many": (c_int, [_P, c_int, _P]),
    "dh_block_act_bytes": (c_int64, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "dh_block_act_offsets": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, POINTER(c_int64)]),
    "dh_block_bwd_scratch_bytes": (c_int64, [c_int, c_int, c_int]),
    "dh_block_fwd": (c_int, [POINTER(BlockArgs), _P]),
    "dh_block_bwd": (c_int, [POINTER(BlockArgs), _P]),
    "dh_attn_fwd": (c_int, [c_int, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "dh_attn_bwd": (c_int, [c_int, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "dh_attn_varlen_fwd": (c_int, [c_int, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "dh_attn_varlen_bwd": (c_int, [c_int, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "dh_attn_bucketed_fwd": (c_int, [c_int, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "dh_attn_bucketed_bwd": (c_int, [c_int, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "dh_attn_pooled_fwd": (c_int, [c_int, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "dh_attn_pooled_bwd": (c_int, [c_int, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "dh_text_embed_fwd": (c_int, [c_int, _P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "dh_text_embed_bwd": (c_int, [c_int, _P, _P, _P, _P, c_int, c_int, c_int, POINTER(c_int64), c_int, _P]),
    "dh_embed_table_grad_ws_bytes": (c_int64, [c_int, c_int]),
    "dh_embed_table_grad": (c_int, [c_int, _P, _P, _P, c_int, c_int, c_int, _P, c_int64, _P]),
    "dh_text_embed_packed_fwd": (c_int, [c_int, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "dh_packed_pos_grad": (c_int, [c_int, _P, _P, c_int, c_int, c_int, _P, _P]),
    "dh_image_prep_u8": (c_int, [_P, c_int, c_int, c_int, _P, _P, POINTER(c_float), POINTER(c_float), _P, c_int, c_int, c_int, c_int, _P]),
    "dh_image_resized_crop_u8": (c_int, [_P, c_int, c_int, c_int, _P, _P, POINTER(c_float), POINTER(c_float), _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "dh_im2row": (c_int, [c_int, _P, c_int, c_int, _P, c_int, c_int, c_int, c_int, _P]),
    "dh_vit_assemble_fwd": (c_int, [c_int, _P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "dh_vit_assemble_bwd": (c_int, [c_int, _P, _P, _P, c_int, c_int, c_int, _P]),
    "dh_pool_rows_fwd": (c_int, [c_int, _P, _P, _P, c_int, c_int, c_int, _P]),
    "dh_pool_rows_bwd": (c_int, [c_int, _P, _P, _P, c_int, c_int, c_int, _P]),
    "dh_l2norm_fwd": (c_int, [c_int, _P, _P, _P, c_int, c_int, c_float, _P]),
    "dh_l2norm_bwd": (c_int, [c_int, _P, _P, _P, _P, c_int, c_int, c_float, _P]),
    "dh_infonce_ws_bytes": (c_int64, [c_int, c_int, c_int]),
    "dh_infonce_fwd": (c_int, [POINTER(NcePair), c_int, c_int, c_int, c_int, _P, c_int, POINTER(c_int), POINTER(c_int), _P, _P, _P, _P, _P, _P, c_int64, _P]),
    "dh_infonce_bwd": (c_int, [POINTER(NcePair), c_int, c_int, c_int, c_int, _P, c_int, POINTER(c_int), POINTER(c_int), _P, _P, _P, _P]),
    "dh_ce_fused_ws_bytes": (c_int64, [c_int, c_int]),
    "dh_ce_fused_fwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, c_int64, _P]),
    "dh_ce_fused_bwd": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, c_int64, _P]),
    "dh_ce_rows_fwd": (c_int, [_P, c_int64, _P, c_int, c_int, _P, _P, _P, _P, _P]),
    "dh_ce_rows_bwd": (c_int, [_P, c_int64, _P, c_int, c_int, _P, _P, _P, c_int64, _P]),
    "dh_ce_rows_bwd_padded": (c_int, [_P, c_int64, _P, c_int, c_int, _P, _P, _P, c_int, c_int64, c_int, c_int, _P]),
    "dh_bn1d_fwd": (c_int, [c_int, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_float, c_float, c_int, c_int, _P]),
    "dh_bn1d_bwd": (c_int, [c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "dh_conv_rows": (c_int, [c_int, _P, c_int, c_int, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "dh_bn2d_ws_bytes": (c_int64, [c_int, c_int]),
    "dh_bn2d_fwd": (c_int, [c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_float, c_float, c_int, c_int, _P, c_int64, _P]),
    "dh_bn2d_bwd": (c_int, [c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P, c_int64, _P]),
    "dh_bn2d_sums": (c_int, [c_int, c_int, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P, _P, c_int64, _P]),
    "dh_bn2d_fwd_apply": (c_int, [c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_float, c_float, c_int, _P]),
    "dh_bn2d_bwd_apply": (c_int, [c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P, c_int64, _P]),
    "dh_avgpool_fwd": (c_int, [c_int, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "dh_avgpool_bwd": (c_int, [c_int, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "dh_attnpool_tokens_fwd": (c_int, [c_int, _P, _P, _P, c_int, c_int, c_int, _P]),
    "dh_attnpool_tokens_bwd": (c_int, [c_int, _P, _P, _P, c_int, c_int, c_int, _P]),
    "dh_cos_rows_fwd": (c_int, [c_int, _P, _P, _P, c_int, c_int, _P]),
    "dh_cos_rows_bwd": (c_int, [c_int, _P, _P, _P, _P, c_int, c_int, _P]),
    "dh_nn_bank_ws_bytes": (c_int64, [c_int, c_int]),
    "dh_nn_bank_query": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, _P, c_int64, _P]),
    "dh_nn_bank_enqueue": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "dh_gather_rows": (c_int, [c_int, _P, _P, _P, c_int, c_int, c_int, _P]),
    "dh_scatter_rows_add": (c_int, [c_int, _P, _P, _P, c_int, c_int, _P]),
    "dh_filip_select": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "dh_maxsim_reduce": (c_int, [c_int, _P, c_int64, c_int, c_int, c_int, _P, _P, _P, _P, _P]),
    "dh_maxsim_fused_fwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P]),
    "dh_maxsim_scatter_rows": (c_int, [c_int, _P, _P, _P, _P, c_int64, c_int, c_int, c_int, c_int64, c_int64, _P]),
    "dh_maxsim_scatter": (c_int, [c_int, _P, _P, _P, _P, c_int64, c_int, c_int, c_int, _P]),
    "dh_adamw": (c_int, [_P, _P, _P, _P, _P, c_int64, c_float, c_float, c_float, c_float, c_float, c_int, c_float, _P]),
    "dh_adamw_segmented": (c_int, [_P, _P, _P, _P, _P, c_int64, _P, _P, _P, c_int, c_float, c_float, c_float, c_int, c_float, _P]),
    "dh_cast": (c_int, [c_int, _P, c_int, _P, c_int64, _P]),
    "dh_comm_unique_id": (c_int, [_P, c_int64]),
    "dh_init": (c_void_p, [c_int, c_int, c_int, _P]),
    "dh_finalize": (c_int, [c_void_p]),
    "dh_ctx_info": (c_int, [c_void_p, POINTER(c_int)]),
    "dh_comm_stream": (c_void_p, [c_void_p]),
    "dh_comm_wait": (c_int, [c_void_p, _P]),
    "dh_allgather_packed": (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_int), c_int, c_int, c_int, _P, _P]),
    "dh_reducescatter_packed": (c_int, [c_void_p, _P, POINTER(c_void_p), POINTER(c_int), c_int, c_int, c_int, _P, _P]),
    "dh_allreduce_bucket": (c_int, [c_void_p, _P, c_int64, _P, _P]),
}

_lib = None


def exported_symbols():
    return sorted(_PROTOS)


def load():
    """dlopen the in-tree library (building it is __graft_entry__.build()'s job)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DeclipHipError(
            "libdeclip_hip.so not found at %s -- run `python -m declip_amd.build` (hipcc, gfx950). "
            "There is no fallback path." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _PROTOS.items():
        try:
            fn = getattr(lib, name)  # AttributeError if the library does not export it
        except AttributeError:
            if os.environ.get("DECLIP_HIP_LIB") and os.environ.get("DH_LIB_ALLOW_MISSING") == "1":
                continue             # an OLDER build as the "before" arm of an A/B run (tools/ab_bench.sh): entry points added since are absent
            raise
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().dh_last_error()
        raise DeclipHipError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))


def dt(t):
    """torch dtype -> DH dtype enum."""
    if t.dtype == torch.float32:
        return DH_F32
    if t.dtype == torch.bfloat16:
        return DH_BF16
    raise DeclipHipError("unsupported dtype %s" % t.dtype)


def torch_dtype(d):
    return torch.float32 if d == DH_F32 else torch.bfloat16


def ptr(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise DeclipHipError("tensor is not on a HIP device (the HIP path has no CPU fallback)")
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def device_info(device=0):
    out = (c_int * 4)()
    check(load().dh_device_info(device, out), "dh_device_info")
    return dict(cus=out[0], clock_khz=out[1], lds_per_cu=out[2], arch=out[3])

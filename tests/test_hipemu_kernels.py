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
    ("test_layernorm", (F32, 77, 512)),
    ("test_layernorm", (BF16, 9, 100)),
    ("test_attention", (BF16, 3, 50, 12, False)),
    ("test_attention", (BF16, 2, 77, 8, True)),
    ("test_attention", (BF16, 2, 33, 1, True)),
    ("test_attention", (F32, 2, 5, 2, False)),
    ("test_attention", (F32, 1, 16, 2, True)),
    ("test_text_embed", (BF16,)),
    ("test_vision_embed", (F32,)),
    ("test_pool_and_l2norm", (BF16,)),
    ("test_infonce", (8, 8, 64, 0)),
    ("test_infonce", (40, 120, 512, 40)),
    ("test_infonce", (33, 99, 768, 66)),
    ("test_ce_rows", ()),
    ("test_adamw_matches_torch", ()),
    ("test_bn1d_groups", (BF16, True)),
    ("test_bn1d_groups", (F32, False)),
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

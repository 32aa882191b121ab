"""Random-shape campaign over the kernels on the host emulation (tests/hipemu): the parametrised `-m gpu` kernel tests and the
ModifiedResNet-op tests are called with random shapes / layouts / dtypes for a time budget; every failure is printed with its
arguments.  Not part of the test suite (minutes); run it after touching a kernel and before the GPU is available.

    python tools/fuzz_kernels_on_host.py [seed] [seconds]

Known non-defects it reports: BatchNorm over 2 rows (dx is ~0 analytically)."""
import os
import random
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
F32, BF16 = torch.float32, torch.bfloat16


def r8(lo, hi):
    return 8 * random.randint(max(1, lo // 8), hi // 8)


def draw():
    kind = random.choice(["gemm", "gemm", "ln", "attn", "nce", "nn", "conv", "bn", "pool", "embgrad", "bn1d", "nce_wide"])
    if kind == "embgrad":       # sort-by-id segmented reduction: few / many distinct ids, d with masked lanes, vocabularies above one scan round
        return "K", "test_embed_table_grad_sorted_segments", (random.randint(20, 3000), 8 * random.randint(1, 100), random.choice([17, 90, 1000, 3000, 49408, 70001]),
                                                               random.choice([F32, BF16]))
    if kind == "bn1d":          # C % 8 == 0: the 16-byte kernels, otherwise the column-per-lane ones; groups with few or many rows
        return "K", "test_bn1d_groups", (random.choice([F32, BF16]), random.random() < 0.5, random.randint(1, 3), random.randint(3, 200),
                                         random.choice([8, 64, 72, 100, 200, 203, 256, 520, 1024]))
    if kind == "nce_wide":      # X staged per pass, dX in 512-column chunks
        b, W = random.randint(16, 50), random.randint(1, 3)
        return "K", "test_infonce", (b, b * W, random.choice([544, 640, 768, 1024, 1280, 1536]), random.randint(0, W - 1) * b)
    if kind == "gemm":
        dtype = random.choice([F32, BF16])
        generic = dtype == BF16 and random.random() < 0.3
        a_km, b_km = random.choice([(False, False), (False, True), (True, True), (True, False)])
        M, N, K = random.randint(1, 300), random.randint(1, 300), random.randint(1, 300)
        if dtype == BF16 and random.random() < 0.7:
            M, N, K = r8(8, 296), r8(8, 296), r8(8, 296)
        return "K", "test_gemm_layouts", (dtype, generic, a_km, b_km, M, N, K)
    if kind == "ln":
        return "K", "test_layernorm", (random.choice([F32, BF16]), random.randint(1, 90), random.choice([8, 36, 64, 100, 128, 256, 384, 512, 640, 768, 1024, 2048]))
    if kind == "attn":
        return "K", "test_attention", (random.choice([F32, BF16]), random.randint(1, 3), random.randint(1, 128), random.randint(1, 4), random.random() < 0.5)
    if kind == "nce":
        b, W = random.randint(16, 70), random.randint(1, 3)
        return "K", "test_infonce", (b, b * W, random.choice([32, 40, 64, 96, 128, 160, 256, 512, 520, 768]), random.randint(0, W - 1) * b)
    if kind == "nn":
        return "K", "test_nn_bank_query_exact", (random.randint(1, 70), random.randint(2, 6000), random.choice([32, 40, 64, 128, 256, 512, 768]))
    if kind == "conv":
        return "R", "test_conv_rows_nhwc", (random.choice([F32, BF16]), random.randint(1, 3), random.randint(1, 20), random.randint(1, 20), 8 * random.randint(1, 9), random.choice([1, 1, 2, 3]))
    if kind == "bn":
        return "R", "test_bn2d_fwd_bwd", (random.choice([F32, BF16]), random.randint(3, 3000), 8 * random.randint(1, 40), random.random() < 0.5, random.random() < 0.5)
    k = random.choice([1, 2, 3, 4])
    return "R", "test_avgpool", (random.choice([F32, BF16]), random.randint(1, 3), k * random.randint(1, 8), k * random.randint(1, 8), 8 * random.randint(1, 6), k)


def main():
    from hipemu_util import emulated_gpu
    import test_gpu_kernels as K
    import test_hipemu_resnet as R
    K.cuda = torch.device("cpu")
    K._poison_lds = lambda ops: None
    random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
    t_end = time.time() + (float(sys.argv[2]) if len(sys.argv) > 2 else 300)
    n = fails = 0
    with emulated_gpu():
        while time.time() < t_end:
            which, name, args = draw()
            n += 1
            try:
                getattr(K if which == "K" else R, name)(*args)
            except Exception as e:      # noqa: BLE001
                fails += 1
                print("FAIL", name, args, repr(e)[:200], flush=True)
    print("ran %d cases, %d failures" % (n, fails))


if __name__ == "__main__":
    main()

"""Run one `-m gpu` test function on the CPU through the host emulation of the kernels (tests/hipemu; DESIGN_HISTORY.md s10).

    python tools/run_gpu_test_on_host.py test_gpu_kernels test_attention "(torch.bfloat16, 2, 77, 8, True)"
    python tools/run_gpu_test_on_host.py test_gpu_clip test_clip_fp32_matches_reference_golden "('clip_tiny',)"
    HIPEMU_TRACE=1 python tools/run_gpu_test_on_host.py ...      # one line per emulated launch (grid, block, LDS, seconds)

Works for tests that do not name the device literally; the inline-ISA GEMM families decline on the host build (the MFMA-builtin
tiles / generic kernel of gemm.hip run instead)."""
import importlib
import os
import sys
import time

import torch  # noqa: F401  (the argument tuple may name torch dtypes)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from hipemu_util import emulated_gpu
    mod = importlib.import_module(sys.argv[1])
    name = sys.argv[2]
    args = eval(sys.argv[3]) if len(sys.argv) > 3 else ()
    if hasattr(mod, "cuda"):
        mod.cuda = torch.device("cpu")
    if hasattr(mod, "_poison_lds"):
        mod._poison_lds = lambda ops: None           # the emulation NaN-poisons dynamic LDS itself
    t0 = time.time()
    with emulated_gpu():
        getattr(mod, name)(*args)
    print("OK %s%s  %.1f s" % (name, args, time.time() - t0))


if __name__ == "__main__":
    main()

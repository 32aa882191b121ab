"""`python bench.py --gpus N` starts its own ranks (the reference starts one process per device itself:
prototype/utils/dist.py:18-24, solver/clip_solver.py:740-764).  CPU tests of the launcher: rendezvous over gloo, the communicator
sanity check, ONE JSON line from rank 0, and the error paths (never an AssertionError)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(argv, env_extra=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p.returncode, lines, p.stderr


def test_bench_self_launch_two_ranks_dry_run_over_gloo():
    rc, lines, err = _run(["--gpus", "2", "--dry-run-launch"], {"DH_DIST_BACKEND": "gloo"})
    assert rc == 0, err[-2000:]
    assert len(lines) == 1, "exactly ONE JSON line from rank 0: %r" % (lines,)
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["rccl_ranks"] == 2 and d["config"]["self_launched"] == 1
    assert len(d["config"]["ranks"]) == 2 and d["config"]["dist_backend"] == "gloo"
    assert "error" not in d


@pytest.mark.parametrize("n", [4, 8])
def test_bench_self_launch_four_and_eight_ranks_dry_run_over_gloo(n):
    """the launch of the driver's scaling points (N = 4, 8), without a model: N processes, one communicator of N ranks, ONE line"""
    rc, lines, err = _run(["--gpus", str(n), "--dry-run-launch"], {"DH_DIST_BACKEND": "gloo"})
    assert rc == 0, err[-2000:]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and d["config"]["rccl_ranks"] == n and len(d["config"]["ranks"]) == n and "error" not in d


def test_bench_self_launch_without_enough_devices_is_a_json_error_not_an_assertion():
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 64:
        pytest.skip("a box with 64 devices")
    rc, lines, err = _run(["--gpus", "64", "--steps", "1", "--warmup", "0"], {"DH_DIST_BACKEND": "nccl"})
    assert rc == 3
    assert len(lines) == 1 and "error" in json.loads(lines[0])
    assert "AssertionError" not in err and "Traceback" not in err


def test_bench_world_size_mismatch_is_a_json_error():
    rc, lines, err = _run(["--gpus", "2", "--dry-run-launch"], {"WORLD_SIZE": "1", "RANK": "0", "DH_DIST_BACKEND": "gloo"})
    assert rc == 3
    assert len(lines) == 1 and "error" in json.loads(lines[0])
    assert "AssertionError" not in err


@pytest.mark.gpu
def test_bench_self_launched_two_ranks_share_the_gpu_over_gloo():
    """The whole bench.py (model, data-parallel step, timed region, JSON line) started as `python bench.py --gpus 2` with no launcher
    around it; two gloo ranks share the one GPU of the test box (RCCL wants a device per rank)."""
    rc, lines, err = _run(["--gpus", "2", "--batch", "256", "--steps", "2", "--warmup", "1", "--no-roofline", "--no-cpu-baseline",
                           "--no-loss-delta"], {"DH_DIST_BACKEND": "gloo"}, timeout=600)
    assert rc == 0, err[-3000:]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["rccl_ranks"] == 2 and d["config"]["self_launched"] == 1 and d["value"] > 0
    assert d["config"]["global_batch"] == 512 and len(d["config"]["ranks"]) == 2


def test_bench_self_launch_times_out_instead_of_hanging(tmp_path):
    """a rank that never finishes: the launcher kills the job after DH_BENCH_LAUNCH_TIMEOUT and prints a JSON error (exit 4)"""
    rc, lines, err = _run(["--gpus", "2", "--dry-run-launch"], {"DH_DIST_BACKEND": "gloo", "DH_BENCH_LAUNCH_TIMEOUT": "0.0"})
    assert rc == 4
    assert len(lines) >= 1 and "timed out" in json.loads(lines[-1])["error"]

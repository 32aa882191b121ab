"""bench.py on a multi-GPU rank's configuration, on ONE GPU (DH_DIST_FORCE=1: an RCCL process group of one rank, every collective of the
W > 1 step issued): since round 5 that configuration replays the CAPTURED step by default, and it has two safety nets for the first
real multi-GPU run -- a capture that raises falls back to the eager step, a warm-up that hangs re-executes the rank with --graph 0.
These tests run the default and force the second net."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(extra_env, *args, timeout=600):
    env = dict(os.environ, DH_DIST_FORCE="1", MASTER_ADDR="127.0.0.1", **extra_env)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "DH_BENCH_GRAPH_FALLBACK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--batch", "256", "--no-cpu-baseline",
                        "--no-loss-delta", "--no-roofline"] + list(args), env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert p.returncode == 0, (p.returncode, p.stderr[-2000:])
    return json.loads(p.stdout.strip().splitlines()[-1]), p.stderr


def test_one_rank_rccl_group_replays_the_captured_step_by_default_and_explains_its_communication():
    d, _ = _bench({})
    assert d["config"]["one_rank_rccl_group"] == 1 and d["config"]["dist_backend"] == "nccl"
    assert d["config"]["step_graph"] == 1 and d["graph_fallback"] is None, (d["config"]["step_graph"], d["graph_fallback"])
    assert d["value"] > 0 and len(d["per_rank_ms"]) == 1
    # the attribution fields of a --gpus N line (a one-rank group issues the same collectives)
    assert d["allreduce_exposed_ms"] >= 0.0 and d["allgather_ms"] > 0.0 and d["allgathers_per_step"] >= 1
    assert sum(d["bucket_mb"]) > 500.0 and max(d["bucket_mb"]) >= d["bucket_mb_configured"] * 0.9       # the 605 MB of fp32 gradients, in buckets


def test_a_hanging_warm_up_of_the_captured_step_reexecutes_the_rank_eagerly():
    d, err = _bench({"DH_BENCH_TEST_HANG": "1", "DH_GRAPH_WATCHDOG_S": "45"})
    assert "watchdog" in err
    assert d["graph_fallback"] and d["graph_fallback"].startswith("watchdog"), d["graph_fallback"]
    assert d["config"]["step_graph"] == 0 and d["value"] > 0


def test_the_default_line_with_its_roofline_pass_runs_end_to_end():
    """`python bench.py` as the driver runs it, roofline pass included (per-op composition with event brackets around every GEMM and
    every grouped weight-gradient launch): the instrumented wrappers must follow the ops' signatures -- round 5 added `first_touch`
    to gemm_dw_group and the first profiling session found the wrapper without it."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "DH_DIST_FORCE")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-loss-delta"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines                       # ONE line on stdout, nothing else
    d = json.loads(lines[0])
    r = d["roofline"]
    assert d["value"] > 0 and d["config"]["step_graph"] == 1 and d["graph_fallback"] is None
    assert r["bound"] == "mfma" and 0.2 < r["frac"] < 1.0 and r["launches_per_step"] > 150 and r["achieved"] > 500
    assert r["traffic"] is None or (r["traffic_file_sha16"] and len(r["traffic_file_sha16"]) == 16)

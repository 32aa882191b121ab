"""Throughput of the CLIP ViT-B/32 training step (BASELINE.json metric: image-text pairs/sec).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python bench.py --gpus N --steps K --warmup W          # starts its own N ranks (one process per GPU over RCCL), or under a launcher:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = forward + fused InfoNCE + backward + gradient all-reduce + fused AdamW on one synthetic
batch (per-GPU batch 512, bf16, BASELINE.json configs[1]); weak scaling.  Prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

GFLOP_PER_PAIR = 43.87          # CLIP ViT-B/32 fwd+bwd, SURVEY.md s8(d) / BASELINE.md s3
GFLOP_PER_PAIR_R50 = 53.9       # CLIP ResNet-50 (configs[0]): trunk 10.73 + attention pool 1.28 + text 5.96 fwd, x3 minus the stem's dX


def dense_gflop_per_pair(model, B):
    """Algorithmic fwd + bwd GFLOP per image-text pair as the REFERENCE spends them (dense towers, 2 flop per MAC, conv1 frozen), per
    model family -- SURVEY.md s8(d) / BASELINE.md s3.  `B` = global batch (the token-wise terms of FILIP grow with it).
      clip      vision 8.818 + text 5.960 fwd; bwd 2 x (fwd - conv1)                                        = 43.87
      declip    2 image + 2 text passes, MLM head on the masked rows, NN search, SimSiam MLPs, 12 logits     = 89.9
      slip      3 image + 1 text passes + the SimCLR head 768-4096-4096-256                                  = 96.1
      filip     towers 14.777 + dense mappings 0.040 + token GEMMs 2 * 256 * 16 * B * (49 + 77) fwd; bwd = 2 x (towers - conv1 +
                mappings) + the token GEMMs' backward with the arg-max sparsity (1 of the 16 selected tokens receives a gradient:
                2/16 of their forward)                                                                        = 46.4 at B = 2048
      defilip   declip + the mappings of both views + FOUR token-wise logits (defilip.py:336-339), each as in filip"""
    tok = 2.0 * 256 * 16 * B * (49 + 77) / 1e9
    if model == "clip":
        return GFLOP_PER_PAIR
    if model == "clip_r50":
        return GFLOP_PER_PAIR_R50
    if model == "declip":
        return 89.9
    if model == "slip":
        return 96.1
    if model == "filip":
        return round((14.777 + 0.040 + tok) + 2.0 * (14.777 - 0.2312 + 0.040) + tok * 2.0 / 16.0, 2)
    if model == "defilip":
        return round(89.9 + 2 * 0.040 * 3 + 4 * (tok + tok * 2.0 / 16.0), 2)
    raise ValueError(model)
PEAK_BF16_TFLOPS = 2500.0       # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)


def _host_cpu():
    """(model string, cores this process may use: physical cores of the affinity mask, capped by the container's CPU quota)."""
    from declip_amd import hostinfo
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return model, hostinfo.usable_cores()


def _cpu_sample(cfg, label, batch, steps, use_reference):
    """pairs/s of ONE configuration on the host cores: the reference's own modules (kind "reference": only where /root/reference
    exists, i.e. the build container) or the oracle restatement of them (kind "port": the GPU box), fp32, fwd + bwd + AdamW."""
    from declip_amd import synth
    from oracle import restated
    sd = synth.synth_state(synth.clip_shapes(cfg), seed=0)
    images = synth.synth_images(batch, seed=0)
    ids = synth.synth_tokens(batch, seed=0)
    if use_reference:
        import contextlib
        import io
        from oracle import gen_golden, ref_harness
        ref = ref_harness.load_reference()
        with contextlib.redirect_stdout(io.StringIO()):
            model = gen_golden.build_ref_clip(ref, cfg, use_allgather=False)
            model.load_state_dict(sd, strict=True)
            model.train()
        gen_golden.patch_tokenize(model.encode_text, {i: ids[i] for i in range(batch)})
        crit = ref.modules["prototype.loss_functions.loss"].ClipInfoCELoss()
        opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-4, betas=(0.9, 0.98), weight_decay=0.1)
        feed = {"images": images, "captions": [[i] for i in range(batch)]}

        def one():
            opt.zero_grad()
            li, lt = model(feed)
            loss, _ = crit(li, lt)
            loss.backward()
            opt.step()
    else:
        frozen = set() if cfg.get("vision") == "resnet" else {"visual.conv1.weight"}
        for k, v in sd.items():
            if v.dtype.is_floating_point and "running_" not in k:
                v.requires_grad_(k not in frozen)
        opt = torch.optim.AdamW([v for v in sd.values() if v.requires_grad], lr=1e-4, betas=(0.9, 0.98), weight_decay=0.1)

        def one():
            opt.zero_grad()
            total, _, _, _ = restated.clip_step_loss(images, ids, sd, cfg, 1)
            total.backward()
            opt.step()
    times = []
    for _ in range(steps + 1):
        t0 = time.time()
        one()
        times.append(time.time() - t0)
    dt = sorted(times[1:])[len(times[1:]) // 2]
    return dict(config=label, value=round(batch / dt, 3), unit="pairs/s", batch=batch, timed_steps=steps, s_per_step=round(dt, 3))


def cpu_baseline(batch=32, steps=3, r50=True):
    """the reference's CPU path beside the GPU number -- CLIP ViT-B/32 (the metric's model) and CLIP ResNet-50
    (BASELINE.json configs[0], the reference's own CPU-runnable case), batch 32, fp32, 1 warm-up + 3 timed steps (median), torch
    threads = the cores this process may use (physical cores, capped by the container's CPU quota: hostinfo.usable_cores).  Bounded: ~10-30 s of CPU work on the GPU box's host."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from declip_amd import synth
    from oracle import ref_harness
    model, cores = _host_cpu()
    prev = torch.get_num_threads()
    torch.set_num_threads(cores)
    use_ref = ref_harness.reference_available()
    try:
        main_s = _cpu_sample(synth.VITB32, "CLIP ViT-B/32", batch, steps, use_ref)
        also = [_cpu_sample(synth.R50, "CLIP ResNet-50 (BASELINE.json configs[0])", batch, steps, use_ref)] if r50 else []
    finally:
        torch.set_num_threads(prev)
    return dict(value=main_s["value"], unit="pairs/s", cores=cores, cpu=model, kind="reference" if use_ref else "port",
                sample="%s fp32 fwd+bwd+AdamW, batch %d, 1 warm-up + %d timed steps (median), torch CPU, %d threads (= usable cores: physical, capped by the cgroup CPU quota); "
                       "%s" % (main_s["config"], batch, steps, cores,
                               "the unmodified reference's modules (oracle/ref_harness.py)" if use_ref else
                               "oracle/restated.py, the CPU restatement of the reference (the reference tree is not on this box)"),
                s_per_step=main_s["s_per_step"], also=also)


def loss_delta_vs_cpu_ref(dev):
    """BASELINE.json metric, second half ("loss delta vs CPU ref"): the engine's loss on the inputs + seeded weights of the fixture
    tests/golden/clip_vitb32_b256.pt against the loss the UNMODIFIED reference computed for them on the CPU in fp32 (fixture
    generated by oracle/gen_golden.py; b = 256 = the smallest batch whose tower GEMMs all run on the benchmarked gemm_v4 kernel)."""
    from declip_amd import synth
    from declip_amd.loss import ClipInfoCELoss
    from declip_amd.testing import build_clip
    path = os.path.join(ROOT, "tests", "golden", "clip_vitb32_b256.pt")
    if not os.path.exists(path):
        return None
    g = torch.load(path, weights_only=False)
    cfg, b, seed = g["cfg"], g["b"], g["seed"]
    images = synth.synth_images(b, res=cfg["res"], seed=seed).to(dev)
    ids = synth.synth_tokens(b, ctx=cfg["ctx"], seed=seed, vocab=cfg["vocab"]).to(dev)
    out = dict(reference_loss=round(g["loss"], 6), batch=b, source="tests/golden/clip_vitb32_b256.pt (reference CPU fp32, oracle/gen_golden.py)")
    for dtype in ("fp32", "bf16"):
        model = build_clip(cfg, dtype=dtype, seed=seed)
        with torch.no_grad():
            li, lt = model({"images": images, "captions": ids})
            loss, _ = ClipInfoCELoss()(li, lt)
        val = float(loss)
        out[dtype] = dict(loss=round(val, 6), rel_delta=float("%.3e" % (abs(val - g["loss"]) / abs(g["loss"]))))
        del model
    torch.cuda.empty_cache()
    return out


# The contract is ONE JSON line on stdout.  Libraries underneath print there too (RCCL 2.26 writes a five-line version banner to
# stdout when its first communicator comes up -- AFTER our line in the pipe, because C stdio is flushed at exit), so a rank keeps the
# real stdout for `_emit` alone and points file descriptor 1 at stderr for everybody else (`_claim_stdout`).
_JSON_FD = None


def _claim_stdout():
    global _JSON_FD
    if _JSON_FD is None:
        sys.stdout.flush()
        _JSON_FD = os.dup(1)
        os.dup2(2, 1)


def _emit(obj):
    line = (json.dumps(obj) + "\n").encode()
    if _JSON_FD is None:
        sys.stdout.write(line.decode())
        sys.stdout.flush()
    else:
        os.write(_JSON_FD, line)


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(n):
    """`python bench.py --gpus N` without a launcher around it: start one process per GPU ourselves -- the reference starts its
    ranks the same way, one process per device with the rank in the environment (prototype/utils/dist.py:18-24,
    solver/clip_solver.py:740-764).  Rank r gets RANK = LOCAL_RANK = r, WORLD_SIZE = N, MASTER_ADDR = 127.0.0.1 and a free port;
    rank 0's stdout (the ONE JSON line) is passed through, the other ranks print nothing.  Fewer devices than ranks (RCCL wants
    one device per rank; DH_DIST_BACKEND=gloo lets ranks share a device for tests): a JSON `error` line and exit code 3 -- never an
    assertion.  Returns the exit code."""
    import subprocess
    backend = os.environ.get("DH_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    dry = "--dry-run-launch" in sys.argv
    if backend == "nccl" and ndev < n and not dry:
        _emit(dict(error="--gpus %d needs %d devices, this box has %d" % (n, n, ndev), n_gpus=n, devices_visible=ndev,
                              hint="DH_DIST_BACKEND=gloo lets several ranks share one device (functional check only)"))
        return 3
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), DH_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on these hosts (RCCL across processes)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    # a rank that hangs (a communicator that never forms, a collective whose peer died without exiting) must not hang the caller
    # for ever: DH_BENCH_LAUNCH_TIMEOUT seconds for the whole job (default 30 minutes), then every rank is killed and the line
    # says so
    deadline = time.time() + float(os.environ.get("DH_BENCH_LAUNCH_TIMEOUT", "1800"))
    try:
        pending = dict(enumerate(procs))
        while pending:
            if time.time() > deadline:
                for q in pending.values():
                    q.kill()
                _emit(dict(error="self-launched job timed out (DH_BENCH_LAUNCH_TIMEOUT)", n_gpus=n, ranks_still_running=sorted(pending)))
                return 4
            for r, p in list(pending.items()):
                code = p.poll()
                if code is None:
                    continue
                del pending[r]
                if code != 0 and rc == 0:
                    rc = code
                    for q in pending.values():      # a rank died: the others would wait in a collective forever
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    if rc not in (0, 3, 4):
        _emit(dict(error="a rank of the self-launched job exited with code %d" % rc, n_gpus=n))
    return rc


def _arm_watchdog(seconds, what):
    """Last line of defence for the DEFAULT multi-GPU configuration (captured step with RCCL collectives in the graph), which has
    never run on more than one rank: if `what` makes no progress for `seconds`, this rank re-executes itself with --graph 0.  A hang
    inside a collective stops every rank, so every rank's watchdog fires and the re-executed ranks meet again in a fresh rendezvous
    (torchrun: the agent's store with the next attempt prefix; self-launched: the next port).  exec closes the device files, so
    the driver tears down the hung queues of the old image.  The line then carries `graph_fallback`.  Returns the timer (cancel())."""
    import threading

    def fire(msg=None):
        msg = msg or "watchdog: %s made no progress in %d s; rank re-executed with --graph 0" % (what, int(seconds))
        sys.stderr.write("bench.py: " + msg + "\n")
        sys.stderr.flush()
        env = dict(os.environ, DH_BENCH_GRAPH_FALLBACK=msg)
        if env.get("TORCHELASTIC_USE_AGENT_STORE") == "True":
            env["TORCHELASTIC_RESTART_COUNT"] = str(int(env.get("TORCHELASTIC_RESTART_COUNT", "0")) + 1)     # fresh key prefix in the agent's store
        elif env.get("MASTER_PORT", "").isdigit():
            env["MASTER_PORT"] = str(int(env["MASTER_PORT"]) + 1)                                              # rank 0 hosts the store itself: next port
        argv, skip = [], False
        for a in sys.argv[1:]:
            if skip:
                skip = False
                continue
            if a == "--graph":
                skip = True
                continue
            if a.startswith("--graph="):
                continue
            argv.append(a)
        # (no closing of other descriptors here: helper threads of the runtime abort the process when their sockets vanish under
        # them, before exec gets to run -- seen on the MI355X.  The device files are opened O_CLOEXEC by the ROCm runtime, so exec
        # itself releases the hung queues; sockets of the old rendezvous that survive are harmless, the new one uses another port.)
        if _JSON_FD is not None:
            os.dup2(_JSON_FD, 1)                  # the new image claims the real stdout again
        os.execve(sys.executable, [sys.executable, os.path.abspath(__file__)] + argv + ["--graph", "0"], env)
    t = threading.Timer(seconds, fire)
    t.daemon = True
    t.reexec = fire           # the same way out for a captured step that RAISES (GraphedStep: invalidated capture / failed first launch)
    t.start()
    return t


def dry_run_launch(args):
    """--dry-run-launch: everything `bench.py --gpus N` does around the timed steps -- process group from the environment, the
    SUM-of-ones check of the communicator, (host, device) of every rank, ONE JSON line from rank 0 -- without a model.  Works on a
    box without a GPU (gloo)."""
    from declip_amd import dist as dh_dist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            _emit(dict(error="--gpus %d but WORLD_SIZE=%d" % (args.gpus, world)))
        return 3
    dh_dist.initialize("nccl")
    import torch.distributed as tdist
    dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    ones = torch.ones(1, device=dev)
    if world > 1:
        tdist.all_reduce(ones)
    ranks = [None] * world
    me = (os.uname().nodename, torch.cuda.current_device() if torch.cuda.is_available() else -1)
    if world > 1:
        tdist.all_gather_object(ranks, me)
    else:
        ranks = [me]
    ok = int(round(float(ones))) == world
    if rank == 0:
        _emit(dict(metric="image-text pairs/sec CLIP ViT-B/32", value=0.0, unit="pairs/s", n_gpus=world, steps=0, warmup=0,
                              dry_run=True, scaling="weak", higher_is_better=True,
                              config=dict(rccl_ranks=int(round(float(ones))), dist_backend=(tdist.get_backend() if world > 1 else None),
                                          ranks=[list(x) for x in ranks], self_launched=int(os.environ.get("DH_BENCH_SELF_LAUNCHED", "0")))),
                         **({} if ok else dict(error="communicator does not match the launch")))
    if world > 1:
        dh_dist.barrier()
        tdist.destroy_process_group()
    return 0 if ok else 3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default 512; clip_r50: 32, the configs[0] batch)")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--model", default="clip", choices=["clip", "declip", "slip", "filip", "defilip", "clip_r50"],
                    help="clip = BASELINE.json metric; declip / slip / filip = configs[2] / [3] / [4] at their per-GPU batches (512 / 512 / 256); "
                         "clip_r50 = configs[0] (CLIP ResNet-50, batch 32; add --dtype fp32)")
    ap.add_argument("--text-packed", choices=["0", "1", "2"], default=None,
                    help="text tower on the caption rows up to <|endoftext|> only (DESIGN_HISTORY.md s11; 1 = variable-length attention, "
                         "2 = attention via the dense layout, 0 = padded as the reference computes them); default: DH_TEXT_PACKED, else 1")
    ap.add_argument("--pooled-last", choices=["0", "1"], default=None,
                    help="last block of each tower for the pooled row only (DESIGN_HISTORY.md s12); default: DH_POOLED_LAST, else 1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-loss-delta", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--graph", choices=["0", "1"], default=None,
                    help="capture forward + loss + backward of the step in ONE HIP graph (declip_amd/graph.py), the fused AdamW stays "
                         "a separate launch; default: DH_STEP_GRAPH, else 1 on one GPU for the models whose graph == eager test gates it (clip, clip_r50, "
                         "declip, defilip, filip), 0 for slip (no gain) and for multi-GPU runs (1 there captures the RCCL collectives with the step)")
    ap.add_argument("--pipeline", choices=["0", "1"], default="0",
                    help="1: the timed steps take their batches from the input pipeline instead of one resident batch -- decoded uint8 images at "
                         "their source size + crop boxes + caption STRINGS through declip_amd.prefetch.DataPrefetcher (BPE on its worker "
                         "thread, pinned upload on a copy stream, resize / mirror / normalise on the GPU; clip_solver.py:30-63, "
                         "imagenet_dataloader.py:36-47).  Graphed like the resident step: one captured graph per padded packed row count (graph.GraphedStep(key=...)).  clip only.")
    ap.add_argument("--dry-run-launch", action="store_true",
                    help="launch check only: rendezvous, communicator sanity check and the JSON line, no model and no timed steps "
                         "(runs on a box without a GPU over gloo: tests/test_bench_launch.py)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))         # `python bench.py --gpus N`: this process becomes the launcher of N ranks
    _claim_stdout()                              # from here on stdout carries the JSON line and nothing else
    if args.dry_run_launch:
        sys.exit(dry_run_launch(args))

    if args.text_packed is not None:
        os.environ["DH_TEXT_PACKED"] = args.text_packed
    if args.pooled_last is not None:
        os.environ["DH_POOLED_LAST"] = args.pooled_last
    from declip_amd import dist as dh_dist
    from declip_amd import engine as eng_mod
    from declip_amd import hostinfo, ops, synth
    host_threads = hostinfo.limit_host_threads()      # the container's CPU quota, not the node's visible cores (hostinfo.py)
    from declip_amd.loss import ClipInfoCELoss
    from declip_amd.optim import build_adamw
    from declip_amd.testing import build_clip, build_declip, declip_batch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    # DH_DIST_FORCE=1 on one GPU: the step of a multi-GPU rank (RCCL process group of ONE rank, packed all-gather + reduce-scatter,
    # bucketed all-reduce inside backward, dynamic tile distribution, eager) -- everything of the W > 1 path but a second rank
    forced = world == 1 and os.environ.get("DH_DIST_FORCE") == "1"
    if forced:
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        os.environ.setdefault("DH_V4_DYNAMIC", "1")
    if world > 1 or forced:
        dh_dist.initialize("nccl")
    else:
        torch.cuda.set_device(0)
    if world != args.gpus:                    # (a launcher that set WORLD_SIZE to something else)
        if rank == 0:
            _emit(dict(error="--gpus %d but WORLD_SIZE=%d" % (args.gpus, world)))
        sys.exit(3)
    dev = torch.device("cuda", torch.cuda.current_device())

    torch.manual_seed(1234 + rank)            # random-init weights, reproducible from run to run (the loss in the line is then too)
    cfg = synth.R50 if args.model == "clip_r50" else synth.FILIP_VITB32 if args.model == "filip" else synth.VITB32
    b = args.batch if args.batch is not None else {"clip_r50": 32, "filip": 256, "defilip": 256}.get(args.model, 512)
    crit = ClipInfoCELoss()
    if args.model in ("clip", "clip_r50"):
        model = build_clip(cfg, dtype=args.dtype, use_allgather=(world > 1 or forced), seed=0, load_synth=False)
        images = synth.synth_images(b, seed=rank).to(dev)
        ids = synth.synth_tokens(b, seed=rank).to(dev)
        batch = {"images": images, "captions": ids}
    elif args.model == "slip":
        from declip_amd.loss import NT_Xent_gather
        from declip_amd.steps import slip_loss
        from declip_amd.testing import build_slip, slip_batch
        model = build_slip(cfg, dtype=args.dtype, seed=0, load_synth=False)
        batch = slip_batch(cfg, b, seed=rank, device=dev)
        simclr_crit = NT_Xent_gather(b)
    elif args.model == "filip":
        from declip_amd.steps import filip_loss
        from declip_amd.testing import build_filip, filip_batch
        model = build_filip(cfg, dtype=args.dtype, seed=0, load_synth=False)
        batch = filip_batch(cfg, b, seed=rank, device=dev)
    elif args.model == "defilip":
        from declip_amd.heads import SimsiamLoss
        from declip_amd.steps import DEFILIP_WEIGHTS, declip_loss
        from declip_amd.testing import build_defilip, defilip_batch
        model = build_defilip(cfg, dtype=args.dtype, seed=0, nn_size=65536, load_synth=False)
        batch = defilip_batch(cfg, b, seed=rank, device=dev)
        sim_crit = SimsiamLoss()
    else:
        from declip_amd.heads import SimsiamLoss
        from declip_amd.steps import declip_loss
        model = build_declip(cfg, dtype=args.dtype, seed=0, nn_size=65536, load_synth=False)
        batch = declip_batch(cfg, b, seed=rank, device=dev)
        sim_crit = SimsiamLoss()
    wrapped = dh_dist.DistModule(model, sync=False)
    opt = build_adamw(model, lr=1e-4, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.1)

    # default: on for one GPU where a graph == eager test gates it (tests/test_gpu_graph.py: clip, clip_r50, declip, defilip, filip);
    # multi-GPU runs capture the RCCL collectives with the step only when asked to (--graph 1 / DH_STEP_GRAPH=1; the capture of a
    # one-rank RCCL step is tested in tests/test_gpu_dist.py, W > 1 has not run anywhere yet) and never over a gloo group
    # Round 5: a multi-GPU rank runs the captured step by DEFAULT too (RCCL collectives captured with it; the one-rank capture is tested in
    # tests/test_gpu_dist.py, DH_DIST_FORCE=1 runs it here) -- with two safety nets, because W > 1 has never executed anywhere: a capture
    # that raises falls back to the eager step on EVERY rank (GraphedStep(fallback=True, agree=...)), and a warm-up that makes no progress
    # (a collective hanging inside a replay) re-executes the rank with --graph 0 (`_arm_watchdog`); either way the line says so in
    # `graph_fallback`.  Never over a gloo group (host-side collectives).
    graph_default = "1" if args.model in ("clip", "clip_r50", "filip", "declip", "defilip", "slip") else "0"
    use_graph = (args.graph if args.graph is not None else os.environ.get("DH_STEP_GRAPH", graph_default)) == "1"
    if world > 1 and torch.distributed.get_backend() != "nccl":
        use_graph = False
    graph_fallback = os.environ.get("DH_BENCH_GRAPH_FALLBACK")        # set by a watchdog re-execution of this rank
    # sanity of a multi-GPU launch before anything is timed: every rank of the launch is in the communicator (a SUM all-reduce of
    # ones over RCCL must count `world` ranks) and sits on its own device
    rccl_ranks = 1
    devs = [(os.uname().nodename, torch.cuda.current_device())]
    if world > 1:
        ones = torch.ones(1, device=dev)
        torch.distributed.all_reduce(ones)
        rccl_ranks = int(round(float(ones)))
        devs = [None] * world
        torch.distributed.all_gather_object(devs, (os.uname().nodename, torch.cuda.current_device()))
        if rccl_ranks != world or torch.distributed.get_world_size() != world or (torch.distributed.get_backend() == "nccl" and len(set(devs)) != world):
            if rank == 0:
                _emit(dict(error="communicator does not match the launch", world=world, rccl_ranks=rccl_ranks,
                                      group_size=torch.distributed.get_world_size(), devices=devs))
            sys.exit(3)

    def fwd_bwd():
        if args.model in ("clip", "clip_r50"):
            li, lt = wrapped(batch)
            loss, _ = crit(li, lt)
            loss = loss / world                  # clip_solver.py:418
        elif args.model == "slip":
            loss = slip_loss(wrapped, batch, crit, simclr_crit, None, world_size=world, with_accuracy=False)["loss"]
        elif args.model == "filip":
            loss = filip_loss(wrapped, batch, crit, world_size=world, with_accuracy=False)["loss"]
        elif args.model == "defilip":
            loss = declip_loss(wrapped, batch, crit, sim_crit, None, weights=DEFILIP_WEIGHTS, world_size=world, with_accuracy=False)["loss"]
        else:
            loss = declip_loss(wrapped, batch, crit, sim_crit, None, world_size=world, with_accuracy=False)["loss"]
        loss.backward()                          # gradient all-reduce overlaps inside (dist.FlatReducer)
        return loss.detach()

    pipeline = None
    if args.pipeline == "1":
        assert args.model == "clip", "--pipeline: the CLIP intake (one view, one caption per pair)"
        import itertools
        import tempfile
        from declip_amd.bpe import NativeTokenizer
        from declip_amd.prefetch import DataPrefetcher
        pool = synth.synth_decoded_batches(b, n_batches=6, seed=rank)
        tok = NativeTokenizer(synth.synthetic_bpe_file(os.path.join(tempfile.gettempdir(), "dh_synthetic_bpe.txt.gz")))
        # W > 1: the padded packed row count of every batch is the MAX over the ranks (dist.RowsSync on the prefetcher's worker thread),
        # so that all ranks key, capture and replay their step graphs in lock-step
        pipeline = DataPrefetcher(itertools.cycle(pool), dev, tokenizer=tok, context_length=77, image_size=224,
                                  rows_sync=dh_dist.RowsSync() if (world > 1 or forced) else None)

    from declip_amd.graph import GraphedStep
    graph_key = None
    if pipeline is not None and use_graph:
        # captions of varying length under a captured step: the batch lives in static buffers, and the graphs are keyed by the
        # padded packed row count of the captions in them (engine.packed_key: the one thing besides the buffer contents that
        # the step's launches depend on; the valid row count is read on the device) -- one graph per 256-row bucket, LRU
        packed_text = eng_mod.text_packed_mode() == 1 and args.dtype == "bf16"
        if packed_text or eng_mod.text_packed_mode() == 0:
            static_images, static_ids = torch.empty_like(batch["images"]), torch.empty_like(batch["captions"])
            batch["images"], batch["captions"] = static_images, static_ids
            if packed_text:
                graph_key = lambda: eng_mod.packed_key(static_ids, torch.bfloat16)      # noqa: E731
        else:
            use_graph = False
    dist_on = world > 1 or forced

    def agree(ok):
        if not dist_on:
            return ok
        flag = torch.tensor([1.0 if ok else 0.0], device=dev)
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
        return bool(float(flag) > 0.5)
    graphed = GraphedStep(fwd_bwd, warmup=2, enabled=use_graph, modules=(wrapped,), key=graph_key, fallback=dist_on, agree=agree if dist_on else None)
    watchdog = _arm_watchdog(float(os.environ.get("DH_GRAPH_WATCHDOG_S", "240")), "the warm-up of the captured multi-GPU step") if (dist_on and use_graph) else None

    def step():
        if pipeline is not None:
            nxt = pipeline.next()                # the caller's stream waits for the copy stream's event: no host synchronisation
            if use_graph:
                static_images.copy_(nxt["images"])
                static_ids.copy_(nxt["captions"])
                tag = getattr(nxt["captions"], "_dh_rows", None)
                if tag is not None:
                    pad = getattr(nxt["captions"], "_dh_rows_pad", None)
                    eng_mod.set_rows_tag(static_ids, tag[1], rows_pad=None if pad is None else pad[1])      # the host-side row count the prefetcher's worker took (+ the job-wide padded count)
            else:
                batch["images"], batch["captions"] = nxt["images"], nxt["captions"]
        opt.zero_grad()
        loss = graphed()
        wrapped.sync_gradients()
        scales = [model.logit_scale] + ([model.logit_scale_dense] if hasattr(model, "logit_scale_dense") else [])
        for p_ in scales:
            p_.data.clamp_(3, 6)                 # grad_clip: logit_scale_param_value (config.yaml:20-23; filip_solver.py:646,661)
        opt.step()
        for p_ in scales:
            p_.data.clamp_(3, 6)
        return loss

    try:
        for _ in range(args.warmup):
            step()
        while use_graph and graphed.enabled and graphed.graph is None:
            step()             # fewer warm-up steps than the capture needs (2 eager + 1 captured): the capture must not land in the timed region
    except RuntimeError as e:
        # a data-parallel rank whose captured step cannot be used and cannot be replaced by the eager step IN THIS PROCESS (GraphedStep:
        # the capture was invalidated, or the first launch failed after the peers entered theirs): the same way out as a hang --
        # this rank starts over with --graph 0; its peers, stuck in a collective without it, follow through their watchdogs
        if watchdog is None or "GraphedStep" not in str(e) or graph_fallback:
            raise
        watchdog.cancel()
        watchdog.reexec("captured step unusable (%s); rank re-executed with --graph 0" % str(e).splitlines()[0][:200])
    if watchdog is not None:
        if graphed.enabled and graphed.graph is not None:
            if os.environ.get("DH_BENCH_TEST_HANG") == "1" and not graph_fallback:      # test hook: a replay that never returns (tests/test_gpu_bench_fallback.py)
                time.sleep(3600)
            step()             # two replays, waited for, before the watchdog is disarmed: a collective that hangs in a replay hangs here
            step()
            torch.cuda.synchronize()
        watchdog.cancel()
    if graphed.fallback_reason and not graph_fallback:
        graph_fallback = graphed.fallback_reason
    if use_graph and graph_key is not None:
        for _ in range(12):   # the input pipeline cycles through 6 batches: every row-count bucket among them is captured before the timed region
            step()
    # long-lived objects (model, optimizer tables, cached workspaces) out of the collector's way: a generation-2 pass over
    # them in the middle of a step costs the host up to ~200 ms (seen on the DeCLIP step, tools/declip_steps.py)
    import gc
    gc.collect()
    gc.freeze()
    gc.disable()           # no cyclic-collector pauses inside the timed steps (re-enabled right after; the steps leave no cycles
                           # that matter over tens of iterations) -- a 50 ms host pause is most of a 67 ms DeCLIP step

    def sync():
        if world > 1:
            dh_dist.barrier()
        torch.cuda.synchronize()

    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    host_elapsed = time.perf_counter() - t0      # the host's share: everything enqueued, nothing waited for yet
    sync()
    elapsed = time.perf_counter() - t0
    gc.enable()
    per_rank_ms = [round(elapsed / args.steps * 1e3, 3)]
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        all_t = [torch.zeros_like(t) for _ in range(world)]
        torch.distributed.all_gather(all_t, t)
        per_rank_ms = [round(float(x) / args.steps * 1e3, 3) for x in all_t]
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t)
    step_graph_used = int(use_graph and graphed.enabled and graphed.graph is not None)      # (read before the instrumented passes below switch the graph off)
    ms_per_step = elapsed / args.steps * 1e3
    host_ms_per_step = host_elapsed / args.steps * 1e3
    pairs_per_s = b * world * args.steps / elapsed

    # the host's real work per step: enqueue time of a step into an EMPTY queue (host_ms_per_step above saturates near the step
    # time whenever the host is the faster side -- the launch queue fills and the runtime makes it wait -- so it cannot tell
    # "host-bound" from "GPU-bound with a full queue")
    enq = []
    gc.disable()
    for _ in range(3):
        sync()
        t1 = time.perf_counter()
        step()
        enq.append(time.perf_counter() - t1)
    sync()
    gc.enable()
    host_enqueue_ms = sorted(enq)[1] * 1e3

    # where a multi-GPU step's time goes, so that whatever scaling efficiency comes out can be attributed without another session:
    # three EAGER steps with event pairs around the collectives (dist.TIMING): the feature all-gather and its reduce-scatter backward on
    # the communication stream, and the tail of the gradient all-reduce that the backward pass could not hide (from "backward fully
    # enqueued" to "all buckets reduced" on the compute stream)
    comm_timing = None
    if dist_on:
        was = graphed.enabled
        graphed.enabled = False
        step()
        sync()
        dh_dist.TIMING = {}
        ncomm = 3
        for _ in range(ncomm):
            step()
        sync()
        ts = dh_dist.timing_summary()
        dh_dist.TIMING = None
        graphed.enabled = was
        bb = ts.get("bucket_bytes", [])
        comm_timing = dict(allreduce_exposed_ms=round(ts.get("allreduce_exposed_ms", 0.0) / ncomm, 3),
                           allgather_ms=round(ts.get("allgather_ms", 0.0) / ncomm, 3), allgathers_per_step=ts.get("allgather_count", 0) // ncomm,
                           reduce_scatter_ms=round(ts.get("reduce_scatter_ms", 0.0) / ncomm, 3),
                           bucket_mb=[round(x / 2 ** 20, 1) for x in bb[:len(bb) // ncomm]],
                           bucket_mb_configured=round(wrapped._flat.reducer.bucket_elems * 4 / 2 ** 20, 1),
                           note="event pairs of 3 eager steps after the timed region, rank 0; allreduce_exposed = gradient all-reduce left after the backward pass was enqueued")

    roofline = None
    if not args.no_roofline:
        # dominant kernel = the MFMA GEMM family: bracket every dh_gemm launch of 2 extra steps with
        # events on the launch stream; achieved = algorithmic 2*M*N*K flops / measured kernel time.
        records = []
        gemm_bytes = []
        graphed.enabled = False            # the instrumented steps run eagerly (a graph replay does not pass through ops.gemm)
        orig = ops.gemm

        def timed_gemm(A, B, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = orig(A, B, **kw)
            e1.record()
            M, K = (A.shape[1], A.shape[0]) if kw.get("a_kmajor") else (A.shape[0], A.shape[1])
            N = B.shape[1] if kw.get("b_kmajor") else B.shape[0]
            esz = A.element_size()
            nbytes = (A.numel() + B.numel()) * esz + out.numel() * out.element_size()
            if kw.get("aux") is not None:
                nbytes += kw["aux"].numel() * kw["aux"].element_size()
            if kw.get("residual") is not None:
                nbytes += kw["residual"].numel() * kw["residual"].element_size()
            if kw.get("accumulate"):
                nbytes += out.numel() * out.element_size()          # read-modify-write of the gradient
            gemm_bytes.append(nbytes)
            records.append((e0, e1, 2.0 * M * N * K, A.dtype,
                            (M, N, K, int(bool(kw.get("a_kmajor"))), int(bool(kw.get("b_kmajor"))), int(kw.get("epilogue", 0) or 0),
                             int(kw.get("residual") is not None), int(bool(kw.get("accumulate"))))))
            return out

        orig_group = ops.gemm_dw_group

        def timed_group(problems, ws=None, first_touch=False):
            # the grouped weight gradients of a block: ONE launch (+ one reduce pass) for all its problems
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            orig_group(problems, ws=ws, first_touch=first_touch)
            e1.record()
            fl, nb = 0.0, 0
            for dy, x, gw, gb in problems:
                fl += 2.0 * dy.shape[1] * x.shape[1] * dy.shape[0]
                nb += (dy.numel() + x.numel()) * dy.element_size() + (1 if first_touch else 2) * gw.numel() * 4     # (first touch: the gradient is written, not read-modify-written)
            gemm_bytes.append(nb)
            dy0 = problems[0][0]
            records.append((e0, e1, fl, dy0.dtype, (sum(p[0].shape[1] * p[1].shape[1] for p in problems) // max(problems[0][1].shape[1], 1),
                                                    problems[0][1].shape[1], dy0.shape[0], 1, 1, 0, 0, len(problems))))
            return None

        ops.gemm = timed_gemm
        import declip_amd.engine as eng
        eng.ops.gemm = timed_gemm
        # kernel durations are taken with the towers on ONE stream: with two streams (the timed region above) a GEMM shares
        # the chip with the other tower's launches and its event-bracketed duration measures the overlap, not the kernel
        streams_env = os.environ.get("DH_TOWER_STREAMS")
        os.environ["DH_TOWER_STREAMS"] = "0"
        native_env = os.environ.get("DH_BLOCK_NATIVE")
        os.environ["DH_BLOCK_NATIVE"] = "0"      # per-op composition: the brackets sit around ops.gemm (same kernels as the C-level block calls)
        # one untimed step in this mode first: the text tower now allocates from the main stream's pool, and a hipMalloc
        # between an event pair (ops.gemm allocates its output) would be billed to that GEMM
        ops.gemm = orig
        eng.ops.gemm = orig
        step()
        ops.gemm = timed_gemm
        eng.ops.gemm = timed_gemm
        ops.gemm_dw_group = timed_group
        sync()
        nprof = 2
        # the event-bracketed durations must be GPU time: (1) every profiled step is enqueued BEHIND a ~60 ms spin kernel, so the
        # host (slower here: two event records per GEMM) is a whole step ahead and no bracket contains a wait for it; (2) what an
        # EMPTY bracket measures on this stream (marker-to-marker latency, queued the same way) is subtracted from every bracket.
        # The committed rocprofv3 kernel table of the same command (profiles/r03_clip_kernel_stats.txt) is the cross-check.
        spin = int(0.060 * getattr(torch.cuda.get_device_properties(dev), "clock_rate", 2400000) * 1e3) if hasattr(torch.cuda, "_sleep") else 0
        empty = []
        if spin:
            torch.cuda._sleep(spin)
        for _ in range(33):
            a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a_.record(); b_.record()
            empty.append((a_, b_))
        sync()
        pair_ms = sorted(a_.elapsed_time(b_) for a_, b_ in empty)[len(empty) // 2]
        tp0 = time.perf_counter()
        for _ in range(nprof):
            if spin:
                torch.cuda._sleep(spin)
            step()
        sync()
        tprof = time.perf_counter() - tp0
        ops.gemm = orig
        eng.ops.gemm = orig
        ops.gemm_dw_group = orig_group
        if streams_env is None:
            del os.environ["DH_TOWER_STREAMS"]
        else:
            os.environ["DH_TOWER_STREAMS"] = streams_env
        if native_env is None:
            del os.environ["DH_BLOCK_NATIVE"]
        else:
            os.environ["DH_BLOCK_NATIVE"] = native_env
        flops = sum(r[2] for r in records if r[3] == torch.bfloat16 or args.dtype != "bf16")
        ms_raw = sum(r[0].elapsed_time(r[1]) for r in records if r[3] == torch.bfloat16 or args.dtype != "bf16")
        ms = sum(max(r[0].elapsed_time(r[1]) - pair_ms, 0.0) for r in records if r[3] == torch.bfloat16 or args.dtype != "bf16")
        achieved = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        if os.environ.get("DH_BENCH_GEMM_TABLE") and rank == 0:   # per-shape in-step GEMM times (tuning aid)
            agg = {}
            for r in records:
                c, tms, fl = agg.get(r[4], (0, 0.0, 0.0))
                agg[r[4]] = (c + 1, tms + r[0].elapsed_time(r[1]), fl + r[2])
            with open(os.environ["DH_BENCH_GEMM_TABLE"], "w") as fh:
                fh.write("M N K ta tb epi res acc(>1: problems of a grouped dW launch, M = sum of their out rows) | calls/step  avg_us  TF/s  ms/step\n")
                for k, (c, tms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                    fh.write("%6d %5d %6d %d %d %d %d %d | %4d %8.1f %6.0f %7.3f\n" % (k + (c // nprof, tms / c * 1e3, fl / tms / 1e9, tms / nprof)))
        # HBM-side traffic of the same kernels from the rocprofv3 PMC passes of tools/profile_step.sh (committed summary):
        # per-launch average next to the algorithmic bytes per launch (operands once + outputs once)
        traffic, traffic_source = None, None
        traffic_sha, traffic_head = None, None
        for rnd in ("r06", "r05", "r04", "r03", "r02", "r01"):         # a COMMITTED rocprofv3 --pmc summary of this configuration (tools/profile_step.sh):
            pmc_file = os.path.join(ROOT, "profiles", "%s_pmc_traffic_%s_b%d.json" % (rnd, args.model, b))   # not measured by this run
            if os.path.exists(pmc_file):
                with open(pmc_file) as fh:
                    pm = json.load(fh)
                traffic = round((pm["gemm_read_bytes_per_step"] + pm["gemm_write_bytes_per_step"]) / max(pm["gemm_launches_per_step"], 1), 1)
                traffic_source = "profiles/%s (committed rocprofv3 --pmc passes of this configuration; NOT re-measured by this run)" % os.path.basename(pmc_file)
                import hashlib
                with open(pmc_file, "rb") as fh:
                    traffic_sha = hashlib.sha256(fh.read()).hexdigest()[:16]          # which file was quoted ...
                traffic_head = pm.get("measured_at_head")                              # ... and the commit its passes ran on (a stale figure is visible in the line)
                break
        roofline = dict(bound="mfma", kernel="v4::gemm_v4_kernel family (every tower GEMM of a step: fwd, dX, dW)", achieved=round(achieved, 2),
                        peak=PEAK_BF16_TFLOPS, unit="TFLOP/s", frac=round(achieved / PEAK_BF16_TFLOPS, 4), traffic=traffic,
                        traffic_unit="bytes per launch (PMC: 2*FETCH_SIZE + WRITE_SIZE)", traffic_source=traffic_source,
                        traffic_file_sha16=traffic_sha, traffic_measured_at_head=traffic_head,
                        algorithmic_bytes_per_launch=round(sum(gemm_bytes) / max(len(gemm_bytes), 1), 1),
                        launches_per_step=len(records) // nprof, gemm_ms_per_step=round(ms / nprof, 3),
                        gemm_ms_per_step_raw_brackets=round(ms_raw / nprof, 3), empty_bracket_us=round(pair_ms * 1e3, 2),
                        note="kernel durations: HIP event brackets with both towers on one stream (no co-running launches), every profiled step "
                             "queued behind a 60 ms spin kernel (no host waits inside a bracket), minus the duration of an empty bracket",
                        gemm_gflop_per_step=round(flops / nprof / 1e9, 1),
                        executed_gemm_gflop_per_pair=round(flops / nprof / 1e9 / b, 2),
                        dense_gflop_per_pair=dense_gflop_per_pair(args.model, b * world),
                        # step-level fractions.  `_executed`: the flops the engine runs (packed captions + pooled last block leave
                        # out work that cannot reach the loss) -- THE roofline fraction of the step.  `_dense_equivalent`: the same
                        # throughput priced at the dense flop count the reference spends for the same outputs (a speed-up figure,
                        # not a utilisation)
                        step_mfma_frac_executed=round(pairs_per_s / b * (flops / nprof) / 1e12 / (PEAK_BF16_TFLOPS * world), 4),
                        step_mfma_frac_dense_equivalent=round(pairs_per_s * dense_gflop_per_pair(args.model, b * world) / 1e3 / (PEAK_BF16_TFLOPS * world), 4))

    name = {"clip": "CLIP ViT-B/32", "declip": "DeCLIP ViT-B/32", "slip": "SLIP ViT-B/32", "filip": "FILIP ViT-B/32", "defilip": "DeFILIP ViT-B/32",
            "clip_r50": "CLIP ResNet-50"}[args.model]
    workloads = {
        "clip": "CLIP ViT-B/32 + 12-layer text transformer, InfoNCE, fwd+bwd+grad-allreduce+AdamW; per-GPU batch %d, 224x224 images, "
                "77-token captions, random-init weights" % b,
        "declip": "DeCLIP ViT-B/32 (2 image views + masked/augmented text, 8+4 InfoNCE pairs, SimSiam, NN bank 65536, MLM), "
                  "fwd+bwd+grad-allreduce+AdamW; per-GPU batch %d" % b,
        "slip": "SLIP ViT-B/32 (CLIP on the base view + SimCLR NT-Xent between two augmented views through the 768-4096-4096-256 MLP; "
                "3 image views per pair), fwd+bwd+grad-allreduce+AdamW; per-GPU batch %d (BASELINE.json configs[3])" % b,
        "filip": "FILIP ViT-B/32, embed 768 (global InfoNCE + token-wise max-sim InfoNCE over 49 image x 77 text tokens, top-16 token "
                 "selection; B x L_i x L_t similarity never materialised), fwd+bwd+grad-allreduce+AdamW; per-GPU batch %d (BASELINE.json configs[4])" % b,
        "defilip": "DeFILIP ViT-B/32 (DeCLIP terms + FILIP token-wise max-sim), fwd+bwd+grad-allreduce+AdamW; per-GPU batch %d" % b,
        "clip_r50": "CLIP ModifiedResNet-50 (per-rank BatchNorm, attention pool) + 12-layer text transformer, InfoNCE, "
                    "fwd+bwd+grad-allreduce+AdamW; per-GPU batch %d (BASELINE.json configs[0]), 224x224 images, 77-token captions" % b,
    }
    out = dict(metric="image-text pairs/sec %s" % name, value=round(pairs_per_s, 2), unit="pairs/s", n_gpus=world,
               steps=args.steps, warmup=args.warmup, ms_per_step=round(ms_per_step, 3), host_ms_per_step=round(host_ms_per_step, 3), host_enqueue_ms_empty_queue=round(host_enqueue_ms, 3),
               higher_is_better=True,
               scaling="weak", vs_baseline=None, dtype=args.dtype, data="synthetic" if pipeline is None else "synthetic, through the input pipeline",
               config=dict(workload=workloads[args.model],
                           global_batch=b * world, per_gpu_batch=b, parallelism="dp%d" % world,
                           tower_streams=1 + len(model.__dict__["_flat_store"].side_streams), step_graph=step_graph_used, graphs_captured=graphed.captures,
                           input_pipeline=(None if pipeline is None else
                                           "DataPrefetcher: uint8 canvases 256x320 (source sizes 192-256 x 256-320) + RandomResizedCrop boxes + mirror "
                                           "flags + caption strings; BPE (dh_bpe_encode) and box bookkeeping on 1 worker thread (%d host cores usable), "
                                           "pinned H2D on a copy stream, dh_image_resized_crop_u8 on the GPU" % hostinfo.usable_cores()),
                           host_threads=host_threads,
                           rccl_ranks=rccl_ranks, dist_backend=(torch.distributed.get_backend() if (world > 1 or forced) else None),
                           one_rank_rccl_group=int(forced),
                           ranks=[list(x) for x in devs], self_launched=int(os.environ.get("DH_BENCH_SELF_LAUNCHED", "0")),
                           dynamic_tiles=int(os.environ.get("DH_V4_DYNAMIC", "0")), comm_native=int(dh_dist.native_comm() is not None) if (world > 1 or forced) else 0,
                           native_blocks=int(eng_mod.native_blocks()),
                           text_packed=eng_mod.text_packed_mode(), pooled_last=int(eng_mod.pooled_last_block())),   # captions computed up to <|endoftext|> only; last block for the pooled rows only (same outputs, fewer executed flops: see roofline.executed_gemm_gflop_per_pair)
               loss=round(float(loss.detach()) * world, 5))
    out["peak_mem_gb"] = round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 2)      # torch allocator high-water mark of this rank (activations of a step + parameters + moments + workspaces)
    out["per_rank_ms"] = per_rank_ms
    out["graph_fallback"] = graph_fallback          # None: the step ran as configured (config.step_graph)
    if comm_timing is not None:
        out.update(comm_timing)
    if roofline is not None:
        out["roofline"] = roofline
    if rank == 0 and world == 1 and args.model == "clip" and not args.no_loss_delta:
        out["loss_delta_vs_cpu_ref"] = loss_delta_vs_cpu_ref(dev)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    if pipeline is not None:
        pipeline.close()
    if rank == 0:
        _emit(out)
    if world > 1 or forced:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

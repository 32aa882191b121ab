"""Throughput of the SOLVER's training step (prototype.solver.clip_solver.ClsSolver on the HIP engine, CLIP ViT-B/32, per-GPU batch 512,
synthetic GPU-resident batches with different caption lengths from step to step), with the captured step (engine.step_graph):

    python tools/solver_step_bench.py                  # one process, no process group
    DH_DIST_FORCE=1 python tools/solver_step_bench.py  # the step of a multi-GPU rank on one GPU (one-rank RCCL group, library communicator,
                                                       # rank-uniform graph key through dist.RowsSync)
VERDICT r5 #4b: the second within 1 % of the first."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

steps = int(os.environ.get("SOLVER_BENCH_STEPS", "30"))
forced = os.environ.get("DH_DIST_FORCE") == "1"
if forced:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29641")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    from declip_amd import dist as dd
    dd.initialize("nccl")
from declip_amd.solver import ClsSolver

img = dict(embed_dim=512, layers=12, heads=12, width=768, input_resolution=224, patch_size=32)
txt = dict(embed_dim=512, context_length=77, transformer_width=512, transformer_heads=8, transformer_layers=12, text_encode_type="Transformer",
           bpe_path=None, text_model_utils=dict(random=False, freeze=False), vocab_size=49409)
cfg = dict(model=dict(type="clip_vitb32", kwargs=dict(image_encode=img, text_encode=txt, clip=dict(use_allgather=forced),
                                                      engine=dict(dtype="bf16", step_graph=os.environ.get("SOLVER_BENCH_GRAPH", "1") == "1"))),
           dist=dict(sync=False), grad_clip=dict(type="logit_scale_param_value", value=3, max_value=6),
           optimizer=dict(type="AdamW", kwargs=dict(lr=1e-4, weight_decay=0.1, betas=[0.9, 0.98], amsgrad=False, eps=1e-8),
                          pconfig={k: dict(weight_decay=0) for k in ("bn_w", "bn_b", "ln_w", "ln_b", "bias", "logit_scale")}),
           lr_scheduler=dict(type="Cosine", kwargs=dict(base_lr=1e-4, warmup_lr=1e-3, min_lr=0.0, warmup_steps=3, max_iter=10000)),
           data=dict(type="clip", read_from="fake", batch_size=512, input_size=224),
           saver=dict(print_freq=1000000, save_freq=0, pretrain=dict(auto_resume=False)))
s = ClsSolver(cfg)
s.loader.n = 6                      # six distinct batches (different packed row counts: several graph keys)
s.train(max_steps=14)               # eager warm-up + captures of every key
torch.cuda.synchronize()
t0 = time.perf_counter()
s.train(max_steps=steps)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
g = s.__dict__.get("_graph")
print("solver step: %.3f ms  %.1f pairs/s  forced_group %d  graph %s captures %s replays %s  fallback %s" % (
    dt * 1e3, 512 / dt, int(forced), g is not None, g["step"].captures if g else 0, g["step"].replays if g else 0,
    g["step"].fallback_reason if g else None))

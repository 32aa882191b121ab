"""Host enqueue time vs GPU step time for every model family at its benchmark batch (a host time close to the step time
means a hidden host<->device synchronisation or a launch-bound step)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from declip_amd import dist as dh_dist, steps, synth, testing
from declip_amd.heads import SimsiamLoss
from declip_amd.loss import ClipInfoCELoss, NT_Xent_gather, NTXentLoss
from declip_amd.optim import build_adamw
kinds = sys.argv[1:] or ["slip", "filip", "defilip"]
for kind in kinds:
    b = 256 if kind in ("filip", "defilip") else 512
    cfg = synth.FILIP_VITB32 if hasattr(synth, "FILIP_VITB32") and kind in ("filip", "defilip") else synth.VITB32
    build = getattr(testing, "build_" + kind)
    kw = dict(nn_size=65536) if kind in ("declip", "defilip") else {}
    model = build(cfg, dtype="bf16", seed=0, load_synth=False, **kw)
    batch = getattr(testing, kind + "_batch")(cfg, b, seed=0, device=torch.device("cuda"))
    wrapped = dh_dist.DistModule(model, sync=False)
    opt = build_adamw(model, lr=1e-4, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.1)
    crit, sim = ClipInfoCELoss(), SimsiamLoss()
    def step():
        opt.zero_grad()
        if kind == "slip":
            out = steps.slip_loss(wrapped, batch, crit, NT_Xent_gather(b), None, world_size=1, with_accuracy=False)
        elif kind == "filip":
            out = steps.filip_loss(wrapped, batch, crit, world_size=1, with_accuracy=False)
        else:
            out = steps.declip_loss(wrapped, batch, crit, sim, None, world_size=1, with_accuracy=False)
        out["loss"].backward(); opt.step()
    for _ in range(4): step()
    torch.cuda.synchronize()
    host, tot = [], []
    for _ in range(12):
        t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        host.append(t1 - t0); tot.append(t2 - t0)
    med = lambda v: sorted(v)[len(v) // 2] * 1e3
    print("%-8s b=%d host %.1f ms (max %.1f)  step %.1f ms (max %.1f)  -> %.0f pairs/s unpipelined" % (kind, b, med(host), max(host) * 1e3, med(tot), max(tot) * 1e3, b / med(tot) * 1e3), flush=True)
    del model, opt, wrapped, batch
    torch.cuda.empty_cache()

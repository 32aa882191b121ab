"""Hardware diagnosis of the two tests that failed on the first MI355X run: prints every error figure
instead of asserting.   python tools/diag_hw.py [r50] [crop]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402


def diag_crop():
    import numpy as np
    from declip_amd import augment, ops
    from oracle import restated
    g = torch.Generator().manual_seed(0)
    sizes = [(480, 640), (333, 500), (600, 400), (224, 224), (1080, 720), (256, 341)]
    canvas = torch.randint(0, 256, (len(sizes), 1080, 720, 3), generator=g, dtype=torch.uint8)
    rng = np.random.default_rng(1)
    flip = torch.tensor([0, 1, 0, 1, 1, 0], dtype=torch.uint8)
    for name, params in (("rrc", augment.random_resized_crop_params(sizes, (224, 224), generator=rng)),
                         ("rcc", augment.resize_center_crop_params(sizes, 256, 224))):
        print(name, "params", params.tolist())
        for rnd in (False, True):
            out = ops.image_resized_crop_u8(canvas.cuda(), torch.from_numpy(params).cuda(), (224, 224), flip=flip.cuda(), round_u8=rnd).cpu()
            ref = restated.image_resized_crop_u8(canvas, params, (224, 224), flip=flip, round_u8=rnd)
            d = (out - ref).abs()
            for i in range(len(sizes)):
                di = d[i]
                idx = int(di.argmax())
                c, y, x = idx // (224 * 224), (idx // 224) % 224, idx % 224
                print("  %s round=%d img %d size %s: max %.3e at (c%d,y%d,x%d)  frac>1e-6 %.2e  frac>1e-4 %.2e nan %d" % (
                    name, rnd, i, sizes[i], float(di.max()), c, y, x, float((di > 1e-6).float().mean()), float((di > 1e-4).float().mean()),
                    int(torch.isnan(out[i]).sum())))


def diag_r50():
    from declip_amd import synth
    from oracle_util import oracle_clip_run
    from test_gpu_resnet_intake_packed import run_engine
    cfg, b, seed = synth.R50, 8, 3
    ref = oracle_clip_run(cfg, b, 1, seed, None)
    _, out = run_engine(cfg, b, seed, "fp32")
    print("loss", out["loss"], float(ref["loss"]), abs(out["loss"] - float(ref["loss"])) / abs(float(ref["loss"])))
    li = ref["per_rank"][0][0].detach()
    print("logits max err / max", float((out["logits_i"] - li).abs().max()), float(li.abs().max()))
    rows = []
    for n, r in ref["grads"].items():
        g = out["grads"].get(n)
        if r is None or g is None:
            print("none:", n, r is None, g is None)
            continue
        rn = float(r.norm())
        rows.append((float((g - r).norm()) / max(rn, 1e-30), n, rn, float(g.norm())))
    rows.sort(reverse=True)
    print("worst 40 gradient errors (rel l2, name, |ref|, |got|):")
    for r in rows[:40]:
        print("  %.3e  %-50s %.4e %.4e" % r)
    import statistics
    print("median rel err", statistics.median(r[0] for r in rows), "n", len(rows))
    for n in ("visual.attnpool.c_proj.weight", "visual.layer4.2.conv3.weight", "visual.layer1.0.conv1.weight", "visual.conv1.weight",
              "encode_text.text_projection.weight"):
        print("named", n, [r[0] for r in rows if r[1] == n])


if __name__ == "__main__":
    what = sys.argv[1:] or ["crop", "r50"]
    if "crop" in what:
        diag_crop()
    if "r50" in what:
        diag_r50()

"""Loss composition of the reference solvers, restated once so that the drop-in solvers, bench.py and the
tests share it.  Reference: solver/clip_solver.py:413-430, solver/declip_solver.py:435-533 (loss terms
each divided by world_size, weighted sum from `clip_simsiam_loss_weight`)."""
import torch

from .loss import accuracy

DECLIP_WEIGHTS = dict(clip_loss=0.4, nn_text=0.2, simsiam_loss=0.2, masking_language=0.2)   # yfcc15m_vit_declip/config.yaml:28-32


def clip_loss(model, batch, criterion, world_size=1, with_accuracy=True):
    """clip_solver.py:413-430."""
    li, lt = model(batch)
    loss, target = criterion(li, lt)
    loss = loss / world_size
    out = dict(loss=loss)
    if with_accuracy:
        p1, p5 = accuracy(li, target, topk=(1, 5), criterion=criterion)
        out.update(top1=p1, top5=p5)
    return out


def declip_loss(model, batch, criterion, simsiam_criterion, nt_xent_criterion=None, weights=None, world_size=1,
                image_text_two_view=True, only_image_two_view=False, with_accuracy=True):
    """declip_solver.py:435-533 (default branch: weighted sum, no 'type')."""
    w = dict(DECLIP_WEIGHTS if weights is None else weights)
    o = model(batch, return_dict=True)
    li1, li2, lt1, lt2 = o["logits"]
    a1, a2, at1, at2 = o["logits_aug"]
    p1, p2, z1, z2 = o["simsiam_features"]
    tf, if1, if2 = o["features"]
    l1, target = criterion(li1, lt1)
    acc_src = criterion.last_correct
    l2, _ = criterion(li2, lt2)
    if only_image_two_view:
        clip = (l1 + l2) / 2
    elif image_text_two_view:
        l1a, _ = criterion(a1, at1)
        l2a, _ = criterion(a2, at2)
        clip = (l1 + l2 + l1a + l2a) / 4
    else:
        raise NotImplementedError()                                   # declip_solver.py:447-452
    clip = clip / world_size
    zero = torch.zeros_like(clip)
    mlm = o["text_self_supervised"] / world_size if "text_self_supervised" in o else zero
    if "nn_text_logits" in o:
        n1, n2, n1a, n2a = o["nn_text_logits"]
        nn_loss = (criterion(n1, n1a)[0] + criterion(n2, n2a)[0]) / 2 / world_size
    else:
        nn_loss = zero
    simsiam = simsiam_criterion(p1, z1, p2, z2) / world_size
    parts = dict(clip=clip, mlm=mlm, nn=nn_loss, simsiam=simsiam)
    if nt_xent_criterion is not None:                                  # logged monitor (declip_solver.py:486-488)
        with torch.no_grad():
            parts["convirt"] = (nt_xent_criterion(if1.detach(), tf.detach()) + nt_xent_criterion(if2.detach(), tf.detach())) / world_size
    loss = clip * w.get("clip_loss", 0)
    if w.get("simsiam_loss", 0):
        loss = loss + simsiam * w["simsiam_loss"]
    if w.get("masking_language", 0):
        loss = loss + mlm * w["masking_language"]
    if w.get("nn_text", 0):
        loss = loss + nn_loss * w["nn_text"]
    if "filip" in o:                                                  # defilip_solver.py:462-478,541-542
        f = criterion(*o["filip"])[0]
        if "filip_aug" in o:
            fa = o["filip_aug"]
            f = (f + criterion(fa[0], fa[1])[0] + criterion(fa[2], fa[3])[0] + criterion(fa[4], fa[5])[0]) / 4
        f = f / world_size
        parts["filip"] = f
        if w.get("filip", 0):
            loss = loss + f * w["filip"]
    out = dict(loss=loss, parts=parts, outputs=o)
    if with_accuracy and acc_src is not None:
        _, c1, c5 = acc_src
        n = target.size(0)
        out.update(top1=c1.sum().reshape(1) * (100.0 / n), top5=c5.sum().reshape(1) * (100.0 / n))
    return out


DEFILIP_WEIGHTS = dict(DECLIP_WEIGHTS, filip=0.2)                    # yfcc15m_vit_defilip/config.yaml:28-34


SLIP_WEIGHTS = dict(clip_loss=1, simclr_loss=1)                        # yfcc15m_vit_slip/config.yaml:28-30


def slip_loss(model, batch, criterion, simclr_criterion, nt_xent_criterion=None, weights=None, world_size=1,
              with_accuracy=True):
    """slip_solver.py:438-527 ('slip' model branch)."""
    w = dict(SLIP_WEIGHTS if weights is None else weights)
    o = model(batch, return_dict=True)
    li, lt = o["logits"]
    tf, imf = o["features"]
    clip, target = criterion(li, lt)
    acc_src = criterion.last_correct
    clip = clip / world_size
    s1, g1, s2, g2 = o["sim_features"]
    simclr = simclr_criterion(s1, g1, s2, g2) / world_size
    parts = dict(clip=clip, simclr=simclr)
    if nt_xent_criterion is not None:                                   # logged monitor (slip_solver.py:500)
        with torch.no_grad():
            parts["nt_xent"] = nt_xent_criterion(imf.detach(), tf.detach()) / world_size
    loss = clip * w.get("clip_loss", 0)
    if w.get("simclr_loss", 0):
        loss = loss + simclr * w["simclr_loss"]
    out = dict(loss=loss, parts=parts, outputs=o)
    if with_accuracy and acc_src is not None:
        _, c1, c5 = acc_src
        n = target.size(0)
        out.update(top1=c1.sum().reshape(1) * (100.0 / n), top5=c5.sum().reshape(1) * (100.0 / n))
    return out


FILIP_WEIGHTS = dict(clip_loss=0.0, clip_dense_loss=1.0)               # yfcc15m_vit_filip/config.yaml:32-37


def filip_loss(model, batch, criterion, weights=None, world_size=1, with_accuracy=True):
    """filip_solver.py:435-532: global InfoNCE (weight 0.0 in the shipped config) + dense max-sim InfoNCE."""
    w = dict(FILIP_WEIGHTS if weights is None else weights)
    o = model(batch, return_dict=True)
    li, lt = o["logits"]
    if w.get("clip_loss", 0):
        clip, target = criterion(li, lt)
    else:
        # the shipped config gives the global InfoNCE weight 0.0 (yfcc15m_vit_filip/config.yaml:33): the reference still runs its
        # backward (a gradient of exact zeros into both towers' features); here the term is evaluated for the log only
        with torch.no_grad():
            clip, target = criterion(li, lt)
    acc_src = criterion.last_correct
    clip = clip / world_size
    parts = dict(clip=clip)
    loss = clip * w.get("clip_loss", 0)
    if "dense_logits" in o:
        dl_i, dl_t = o["dense_logits"]
        dense = criterion(dl_i, dl_t)[0] / world_size
        parts["dense"] = dense
        if w.get("clip_dense_loss", 0):
            loss = loss + dense * w["clip_dense_loss"]
    out = dict(loss=loss, parts=parts, outputs=o)
    if with_accuracy and acc_src is not None:
        _, c1, c5 = acc_src
        n = target.size(0)
        out.update(top1=c1.sum().reshape(1) * (100.0 / n), top5=c5.sum().reshape(1) * (100.0 / n))
    return out

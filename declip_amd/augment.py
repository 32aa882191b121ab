"""Crop-box generators for the on-GPU image intake (`ops.image_resized_crop_u8` / `dh_image_resized_crop_u8`).

The reference builds its image augmentation from torchvision transforms on PIL images inside DataLoader workers (or DALI):
`STANDARD_SLIP` = RandomResizedCrop(224, scale=(0.5, 1.0)), `STANDARD_CLIP` = RandomCropMinSize(224), `ONECROP` = Resize(256) +
CenterCrop(224), each followed by ToTensor + Normalize (data/imagenet_dataloader.py:36-47,105-111; data/transforms.py:133-157).
All of them are "take a box of the decoded image, resize it": here the host only draws the BOXES (a few integers per image --
this file, numpy, no pixels touched) and the GPU does the pixels (resize, mirror, normalise in one kernel), so decoded uint8
images go over PCIe once at their source size and no CPU core spends ~2 ms per image in PIL's resize.

Each function returns an int32 array [b, 8] = (x0, y0, w, h, Wf, Hf, ox, oy) per image (include/declip_hip.h).
"""
import math

import numpy as np


def _rng(generator):
    return generator if generator is not None else np.random.default_rng()


def random_resized_crop_params(sizes, out_hw, scale=(0.5, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0), generator=None):
    """torchvision.transforms.RandomResizedCrop.get_params, restated: up to 10 draws of (area fraction ~ U(scale), aspect ~ log-uniform
    (ratio)), accepted when the box fits; else the central box with the aspect ratio clamped into `ratio`.
    sizes: iterable of (h, w) of the decoded images."""
    g = _rng(generator)
    H, W = out_hw
    out = np.zeros((len(sizes), 8), dtype=np.int32)
    log_r = (math.log(ratio[0]), math.log(ratio[1]))
    for n, (h, w) in enumerate(sizes):
        area = float(h) * float(w)
        box = None
        for _ in range(10):
            target = area * g.uniform(scale[0], scale[1])
            aspect = math.exp(g.uniform(log_r[0], log_r[1]))
            cw, ch = int(round(math.sqrt(target * aspect))), int(round(math.sqrt(target / aspect)))
            if 0 < cw <= w and 0 < ch <= h:
                box = (int(g.integers(0, w - cw + 1)), int(g.integers(0, h - ch + 1)), cw, ch)
                break
        if box is None:
            in_ratio = float(w) / float(h)
            if in_ratio < ratio[0]:
                cw, ch = w, int(round(w / ratio[0]))
            elif in_ratio > ratio[1]:
                ch, cw = h, int(round(h * ratio[1]))
            else:
                cw, ch = w, h
            box = ((w - cw) // 2, (h - ch) // 2, cw, ch)
        out[n] = (box[0], box[1], box[2], box[3], W, H, 0, 0)
    return out


def random_crop_min_size_params(sizes, size, generator=None):
    """data/transforms.py:133-157 (RandomCropMinSize, the `STANDARD_CLIP` augmentation): a random SQUARE box spanning the shorter
    side, resized to size x size."""
    g = _rng(generator)
    out = np.zeros((len(sizes), 8), dtype=np.int32)
    for n, (h, w) in enumerate(sizes):
        if w < h:
            box = (0, int(g.integers(0, int(round(h - w)) + 1)), w, w)
        elif w > h:
            box = (int(g.integers(0, int(round(w - h)) + 1)), 0, h, h)
        else:
            box = (0, 0, w, h)
        out[n] = (box[0], box[1], box[2], box[3], size, size, 0, 0)
    return out


def resize_center_crop_params(sizes, resize=256, crop=224):
    """torchvision Resize(resize) (shorter side -> `resize`, the other int(resize * long / short)) + CenterCrop(crop)
    (data/imagenet_dataloader.py:105-111, the zero-shot `ONECROP` pipeline): the whole image is the box, the output window is the
    centre of the resized image."""
    out = np.zeros((len(sizes), 8), dtype=np.int32)
    for n, (h, w) in enumerate(sizes):
        if w <= h:
            Wf, Hf = resize, int(resize * h / w)
        else:
            Hf, Wf = resize, int(resize * w / h)
        if Wf < crop or Hf < crop:
            raise ValueError("image %dx%d is smaller than the crop after Resize(%d)" % (w, h, resize))
        out[n] = (0, 0, w, h, Wf, Hf, int(round((Wf - crop) / 2.0)), int(round((Hf - crop) / 2.0)))
    return out


def make_canvas(images, pinned=False):
    """Stack decoded HWC uint8 images of different sizes on one [b, Hmax, Wmax, 3] canvas (each in its top-left corner) -- the
    `src` of dh_image_resized_crop_u8 -- and return (canvas tensor, [(h, w), ...]).  The padding is never read: every crop box
    lies inside its image.  `pinned` allocates page-locked memory so that the upload is an asynchronous copy."""
    import torch
    sizes = [(int(im.shape[0]), int(im.shape[1])) for im in images]
    H, W = max(h for h, _ in sizes), max(w for _, w in sizes)
    canvas = torch.zeros(len(images), H, W, 3, dtype=torch.uint8, pin_memory=bool(pinned))
    for n, im in enumerate(images):
        t = im if torch.is_tensor(im) else torch.from_numpy(np.ascontiguousarray(im))
        if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] != 3:
            raise ValueError("image %d: expected uint8 [h, w, 3], got %s %s" % (n, t.dtype, tuple(t.shape)))
        canvas[n, :t.shape[0], :t.shape[1]] = t
    return canvas, sizes

"""ORACLE - TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench cpu leg may import this; the product
path must not).

CPU restatement of the reference's training-time data augmentation, `OpenImages.__getitem__` / `_transforms`
(/root/reference/src/helpers/datasets.py:181-270):

    scale ~ U(max(crop / min(H, W), 0.75), max(.., 0.95))                                   datasets.py:251-256
    RandomHorizontalFlip() -> Resize((ceil(scale H), ceil(scale W))) -> RandomCrop(crop) -> ToTensor()   :206-216

The arithmetic lives in two third-party dependencies that are NOT vendored in /root/reference:
  * Pillow (pinned Pillow==8.3.2, requirements.txt:21): `Image.resize(size, BILINEAR)` = libImaging/Resample.c,
    8 bits per channel path: per axis `precompute_coeffs` (double precision: support = max(in/out, 1) x 1.0,
    centre = (x + .5) in/out, window [int(centre - support + .5), int(centre + support + .5)) clipped to the image,
    triangle weights normalised to sum 1), `normalize_coeffs_8bpc` (fixed point, PRECISION_BITS = 32 - 8 - 2 = 22,
    round half away from zero), then a horizontal pass into an 8-bit intermediate and a vertical pass, each
    `clip8((sum + 2^21) >> 22)`.  Restated below in numpy.
  * torchvision (pinned 0.7.0, requirements.txt:38): RandomHorizontalFlip flips iff `torch.rand(1) < 0.5`;
    `Resize(size)` on a PIL image calls `img.resize(size[::-1], BILINEAR)`; RandomCrop.get_params draws
    `i = torch.randint(0, h - th + 1, (1,))`, then `j = torch.randint(0, w - tw + 1, (1,))`; ToTensor = uint8 HWC ->
    float32 CHW / 255.
Parity pin: this file is checked against Pillow itself (installed in this image, 12.2.0: Resample.c's 8-bit bilinear
path is unchanged since 3.x) in tests/test_augment.py, live and through tests/golden/augment_golden.npz.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2          # Resample.c: #define PRECISION_BITS (32 - 8 - 2)
SCALE_MIN, SCALE_MAX = 0.75, 0.95    # datasets.py:20-21


def precompute_coeffs(in_size, out_size):
    """Resample.c precompute_coeffs for the bilinear filter over the whole axis (box = [0, in_size)).
    Returns (bounds[out_size, 2] = (xmin, count), kk[out_size, ksize] int32 fixed-point weights)."""
    scale = filterscale = float(in_size) / float(out_size)
    if filterscale < 1.0:
        filterscale = 1.0
    support = 1.0 * filterscale                    # bilinear: filter support 1.0
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = np.zeros(ksize, dtype=np.float64)
        ww = 0.0
        for x in range(xmax):
            t = (x + xmin - center + 0.5) * ss
            t = -t if t < 0 else t
            v = 1.0 - t if t < 1.0 else 0.0        # bilinear_filter
            w[x] = v
            ww += v
        for x in range(xmax):
            if ww != 0.0:
                w[x] /= ww
        # normalize_coeffs_8bpc: round half away from zero
        for x in range(ksize):
            kk[xx, x] = int(-0.5 + w[x] * (1 << PRECISION_BITS)) if w[x] < 0 else int(0.5 + w[x] * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _clip8(v):
    return np.clip(v >> PRECISION_BITS, 0, 255).astype(np.uint8)


def resize_bilinear_u8(img, out_h, out_w):
    """PIL `Image.fromarray(img).resize((out_w, out_h), BILINEAR)` for uint8 (H, W, C): horizontal pass first (only
    when the width changes), 8-bit intermediate, then the vertical pass (Resample.c ImagingResampleInner)."""
    H, W, C = img.shape
    cur = img
    half = 1 << (PRECISION_BITS - 1)
    if out_w != W:
        b, kk = precompute_coeffs(W, out_w)
        tmp = np.zeros((H, out_w, C), dtype=np.uint8)
        for xx in range(out_w):
            x0, n = b[xx]
            acc = np.full((H, C), half, dtype=np.int64)
            for x in range(n):
                acc += cur[:, x0 + x, :].astype(np.int64) * int(kk[xx, x])
            tmp[:, xx, :] = _clip8(acc)
        cur = tmp
    if out_h != H:
        b, kk = precompute_coeffs(H, out_h)
        tmp = np.zeros((out_h, cur.shape[1], C), dtype=np.uint8)
        for yy in range(out_h):
            y0, n = b[yy]
            acc = np.full((cur.shape[1], C), half, dtype=np.int64)
            for y in range(n):
                acc += cur[y0 + y].astype(np.int64) * int(kk[yy, y])
            tmp[yy] = _clip8(acc)
        cur = tmp
    return cur


def scale_range(H, W, crop, scale_min=SCALE_MIN, scale_max=SCALE_MAX):
    """datasets.py:249-255."""
    shortest = min(H, W)
    lo = max(float(crop) / float(shortest), scale_min)
    hi = max(lo, scale_max)
    return lo, hi


def resized_dims(H, W, scale):
    return math.ceil(scale * H), math.ceil(scale * W)        # datasets.py:210


def augment(img, scale, flip, top, left, crop=256):
    """The deterministic part of datasets.py:206-216 for given random draws: uint8 (H, W, 3) -> float32 (3, crop, crop)
    in [0, 1]."""
    H, W, _ = img.shape
    if flip:
        img = img[:, ::-1, :]
    oh, ow = resized_dims(H, W, scale)
    r = resize_bilinear_u8(np.ascontiguousarray(img), oh, ow)
    c = r[top:top + crop, left:left + crop, :]
    return np.ascontiguousarray(c.transpose(2, 0, 1)).astype(np.float32) / np.float32(255.0)

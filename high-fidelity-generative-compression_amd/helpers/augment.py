"""Training-time augmentation of the reference's OpenImages dataset (src/helpers/datasets.py:181-270) as ONE device
launch per batch: random horizontal flip, random rescale in [0.75, 0.95] (raised so that the shorter side still covers
the crop), random `crop_size` crop, conversion to float32 CHW in [0, 1] (optionally normalised to [-1, 1]).

Host side (this file): the random draws - in the reference's order and from the same generators (numpy for the scale,
datasets.py:256; torch for flip and crop origin, torchvision 0.7 RandomHorizontalFlip / RandomCrop.get_params) - and
Pillow's per-axis fixed-point resampling weights for the crop window only (libImaging/Resample.c precompute_coeffs +
normalize_coeffs_8bpc, double precision, vectorised in numpy).  Device side: csrc/augment.hip gathers every output
pixel from the decoded uint8 image; results are bit-identical to PIL's.

    aug = GpuAugmenter(crop_size=256, device="cuda")
    batch = aug([img0_u8_hwc, img1_u8_hwc, ...])              # (B,3,256,256) float32 on the device
"""
import ctypes
import math

import numpy as np
import torch

from .. import lib

SCALE_MIN, SCALE_MAX = 0.75, 0.95          # datasets.py:20-21
PRECISION_BITS = 22                        # Resample.c: 32 - 8 - 2
KMAX_DEFAULT = 8


class AugParams:
    __slots__ = ("scale", "flip", "top", "left", "out_h", "out_w")

    def __init__(self, scale, flip, top, left, out_h, out_w):
        self.scale, self.flip, self.top, self.left, self.out_h, self.out_w = scale, flip, top, left, out_h, out_w


def draw_params(H, W, crop_size=256, np_random=np.random, generator=None, scale_min=SCALE_MIN, scale_max=SCALE_MAX):
    """One sample's random draws, in the reference's order: scale (datasets.py:251-256, numpy), flip (`torch.rand(1) <
    0.5`), crop row then crop column (`torch.randint(0, h - th + 1, (1,))`)."""
    lo = max(float(crop_size) / float(min(H, W)), scale_min)
    hi = max(lo, scale_max)
    scale = float(np_random.uniform(lo, hi))
    flip = bool(torch.rand(1, generator=generator) < 0.5)
    out_h, out_w = math.ceil(scale * H), math.ceil(scale * W)
    if out_h < crop_size or out_w < crop_size:
        raise ValueError(f"Required crop size {(crop_size, crop_size)} is larger then input image size {(out_h, out_w)}")
    if out_w == crop_size and out_h == crop_size:
        top = left = 0                                           # RandomCrop.get_params short-cut: no draws
    else:
        top = int(torch.randint(0, out_h - crop_size + 1, size=(1,), generator=generator).item())
        left = int(torch.randint(0, out_w - crop_size + 1, size=(1,), generator=generator).item())
    return AugParams(scale, flip, top, left, out_h, out_w)


def window_coeffs(in_size, out_size, first, count, kmax=KMAX_DEFAULT):
    """Pillow's bilinear weights for output positions [first, first+count) of an axis resized in_size -> out_size.
    Returns (bounds int32 [count,2] = (first source index, taps), weights int32 [count,kmax])."""
    scale = float(in_size) / float(out_size)
    filterscale = max(scale, 1.0)
    support = filterscale                                        # bilinear support 1.0
    ksize = int(math.ceil(support)) * 2 + 1
    if ksize > kmax:
        raise ValueError(f"down-scaling by {scale:.3f} needs {ksize} taps per pixel (kmax={kmax})")
    xx = np.arange(first, first + count, dtype=np.float64)
    center = (xx + 0.5) * scale
    xmin = np.maximum((center - support + 0.5).astype(np.int64), 0)          # C's (int) truncation: arguments >= 0
    xmax = np.minimum((center + support + 0.5).astype(np.int64), in_size)
    n = (xmax - xmin).astype(np.int64)
    j = np.arange(kmax, dtype=np.float64)[None, :]
    t = np.abs((j + xmin[:, None] - center[:, None] + 0.5) * (1.0 / filterscale))
    w = np.where(t < 1.0, 1.0 - t, 0.0)
    w = np.where(np.arange(kmax)[None, :] < n[:, None], w, 0.0)
    # ww accumulates left to right in C; summing <= 5 doubles in that order
    ww = np.zeros(count, dtype=np.float64)
    for k in range(kmax):
        ww = ww + w[:, k]
    w = np.where(ww[:, None] != 0.0, w / np.where(ww == 0.0, 1.0, ww)[:, None], w)
    kk = (0.5 + w * float(1 << PRECISION_BITS)).astype(np.int64).astype(np.int32)   # weights are >= 0
    bounds = np.stack([xmin, n], axis=1).astype(np.int32)
    return bounds, kk


class _AugImage(ctypes.Structure):          # hific_aug_image (include/hific_hip.h)
    _fields_ = [("src", ctypes.c_void_p), ("H", ctypes.c_int), ("W", ctypes.c_int), ("flip", ctypes.c_int),
                ("resize_x", ctypes.c_int), ("resize_y", ctypes.c_int), ("top", ctypes.c_int), ("left", ctypes.c_int)]


class GpuAugmenter:
    def __init__(self, crop_size=256, normalize=False, device="cuda", kmax=KMAX_DEFAULT, scale_min=SCALE_MIN,
                 scale_max=SCALE_MAX):
        self.crop, self.normalize, self.kmax = int(crop_size), bool(normalize), int(kmax)
        self.device = torch.device(device)
        self.scale_min, self.scale_max = scale_min, scale_max

    def __call__(self, images, params=None, np_random=np.random, generator=None):
        """images: list of uint8 (H, W, 3) arrays / tensors (host or device).  params: optional list of AugParams
        (default: drawn like the reference).  Returns float32 (B, 3, crop, crop) on the device."""
        B, crop, kmax = len(images), self.crop, self.kmax
        if B == 0:
            return torch.empty((0, 3, crop, crop), dtype=torch.float32, device=self.device)
        dev_imgs, descs = [], (_AugImage * B)()
        xb = np.zeros((B, crop, 2), np.int32); yb = np.zeros((B, crop, 2), np.int32)
        xk = np.zeros((B, crop, kmax), np.int32); yk = np.zeros((B, crop, kmax), np.int32)
        for i, im in enumerate(images):
            t = im if isinstance(im, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(im))
            if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] != 3:
                raise lib.HificError("augmentation input must be uint8 (H, W, 3)")
            H, W = int(t.shape[0]), int(t.shape[1])
            p = params[i] if params is not None else draw_params(H, W, crop, np_random, generator, self.scale_min,
                                                                 self.scale_max)
            if p.top < 0 or p.left < 0 or p.top + crop > p.out_h or p.left + crop > p.out_w:
                raise lib.HificError("crop window outside the resized image")
            t = t.to(self.device, non_blocking=True).contiguous()
            dev_imgs.append(t)
            d = descs[i]
            d.src, d.H, d.W, d.flip, d.top, d.left = t.data_ptr(), H, W, int(p.flip), p.top, p.left
            d.resize_x, d.resize_y = int(p.out_w != W), int(p.out_h != H)          # Pillow skips an unchanged axis
            if d.resize_x:
                xb[i], xk[i] = window_coeffs(W, p.out_w, p.left, crop, kmax)
            if d.resize_y:
                yb[i], yk[i] = window_coeffs(H, p.out_h, p.top, crop, kmax)
        up = lambda a: torch.from_numpy(a).to(self.device, non_blocking=True)
        dd = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(self.device, non_blocking=True)
        xb_d, xk_d, yb_d, yk_d = up(xb), up(xk), up(yb), up(yk)
        out = torch.empty((B, 3, crop, crop), dtype=torch.float32, device=self.device)
        lib.require_gpu(out, dd, xb_d, xk_d, yb_d, yk_d, *dev_imgs)
        lib.call("hific_augment_crop", dd.data_ptr(), xb_d.data_ptr(), xk_d.data_ptr(), yb_d.data_ptr(), yk_d.data_ptr(),
                 B, crop, kmax, int(self.normalize), out.data_ptr(), lib.stream())
        # dev_imgs / tables must outlive the asynchronous launch: stream-ordered frees of the caching allocator do
        out._hific_keepalive = (dev_imgs, dd, xb_d, xk_d, yb_d, yk_d)
        return out

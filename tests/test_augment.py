"""SURVEY section 8 f4 - the data-augmentation path (src/helpers/datasets.py:181-270).

CPU: the oracle's restatement of Pillow's 8-bit bilinear resampler against golden vectors made with Pillow and, where
Pillow is importable, against Pillow live; the product's host logic (random draws in the reference's order, crop-window
weights) against the oracle.  GPU: the HIP kernel against the oracle, bit-exact, on ragged batches."""
import math
import os
import sys
import zlib

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from oracle import augment_oracle as A
import make_augment_golden as G

GOLD = np.load(os.path.join(ROOT, "tests", "golden", "augment_golden.npz"))


def _u8(x):
    """float32 CHW in [0,1] -> uint8 HWC (exact inverse of ToTensor)."""
    return np.rint(np.asarray(x, dtype=np.float64) * 255.0).astype(np.uint8).transpose(1, 2, 0)


def test_oracle_matches_pillow_goldens():
    for i, row in enumerate(GOLD["cases"]):
        H, W, crop, flip, top, left, crc = int(row[0]), int(row[1]), int(row[2]), int(row[4]), int(row[5]), int(row[6]), int(row[7])
        scale = float(row[3])
        got = _u8(A.augment(G.image(100 + i, H, W), scale, flip, top, left, crop))
        assert np.array_equal(got[:8, :8], GOLD[f"patch{i}"]) and np.array_equal(got[-4:, -4:], GOLD[f"last{i}"]), i
        assert zlib.crc32(got.tobytes()) == crc, i


def test_oracle_matches_pillow_live():
    PIL = pytest.importorskip("PIL")
    from PIL import Image
    rng = np.random.default_rng(5)
    for trial in range(10):
        H, W = int(rng.integers(30, 120)), int(rng.integers(30, 120))
        a = G.image(trial, H, W)
        s = float(rng.uniform(0.55, 1.6))
        oh, ow = A.resized_dims(H, W, s)
        ref = np.asarray(Image.fromarray(a).resize((ow, oh), Image.BILINEAR))
        assert np.array_equal(A.resize_bilinear_u8(a, oh, ow), ref), (trial, H, W, s)


def test_scale_range_and_draw_order(hific):
    """datasets.py:249-256: scale_low = max(crop / shortest side, 0.75), scale_high = max(scale_low, 0.95); the
    draws come in the reference's order from numpy (scale) then torch (flip, crop row, crop column)."""
    from hific_amd.helpers import augment
    assert A.scale_range(512, 768, 256) == (0.75, 0.95)
    assert A.scale_range(300, 400, 256) == (256 / 300, 0.95)
    assert A.scale_range(200, 400, 256) == (1.28, 1.28)
    st = np.random.RandomState(3)
    g = torch.Generator().manual_seed(11)
    p = augment.draw_params(300, 400, 256, np_random=st, generator=g)
    st2 = np.random.RandomState(3)
    g2 = torch.Generator().manual_seed(11)
    scale = st2.uniform(256 / 300, 0.95)
    flip = bool(torch.rand(1, generator=g2) < 0.5)
    oh, ow = math.ceil(scale * 300), math.ceil(scale * 400)
    top = int(torch.randint(0, oh - 256 + 1, size=(1,), generator=g2))
    left = int(torch.randint(0, ow - 256 + 1, size=(1,), generator=g2))
    assert (p.scale, p.flip, p.top, p.left, p.out_h, p.out_w) == (scale, flip, top, left, oh, ow)
    # a small image is scaled UP so that its shorter side covers the crop (scale_low = crop / shortest side)
    q = augment.draw_params(100, 180, 256, np_random=np.random.RandomState(0), generator=torch.Generator().manual_seed(0))
    assert q.scale == 2.56 and q.out_h == 256 and q.top == 0 and 0 <= q.left <= q.out_w - 256


def test_window_coeffs_equal_pillow_precompute(hific):
    from hific_amd.helpers import augment
    for in_size, out_size in ((400, 342), (300, 256), (131, 109), (200, 256), (260, 333), (97, 81), (1024, 769)):
        b, kk = A.precompute_coeffs(in_size, out_size)
        for first, count in ((0, min(out_size, 64)), (out_size - 64 if out_size > 64 else 0, min(out_size, 64))):
            wb, wk = augment.window_coeffs(in_size, out_size, first, count)
            assert np.array_equal(wb, b[first:first + count])
            ks = kk.shape[1]
            assert np.array_equal(wk[:, :ks], kk[first:first + count]) and not wk[:, ks:].any()
    with pytest.raises(ValueError):
        augment.window_coeffs(4000, 500, 0, 16)          # 8x down-scaling: more taps than kmax


def test_augmenter_refuses_cpu_and_bad_input(hific):
    from hific_amd.helpers import augment
    from hific_amd import lib
    aug = augment.GpuAugmenter(crop_size=32, device="cpu")
    with pytest.raises(lib.HificError):
        aug([np.zeros((40, 40, 3), np.uint8)])           # no CPU fallback: the kernel is the only implementation
    with pytest.raises(lib.HificError):
        aug([np.zeros((40, 40), np.uint8)])


@pytest.mark.gpu
def test_gpu_augment_bit_exact_ragged_batch(hific, dev):
    """One launch over a ragged batch (the golden cases' sizes and draws) == Pillow's result, bit for bit."""
    from hific_amd.helpers import augment
    by_crop = {}
    for i, row in enumerate(GOLD["cases"]):
        by_crop.setdefault(int(row[2]), []).append((i, row))
    for crop, rows in by_crop.items():
        aug = augment.GpuAugmenter(crop_size=crop, device=dev)
        imgs, params = [], []
        for i, row in rows:
            H, W, scale, flip, top, left = int(row[0]), int(row[1]), float(row[3]), int(row[4]), int(row[5]), int(row[6])
            imgs.append(G.image(100 + i, H, W))
            oh, ow = A.resized_dims(H, W, scale)
            params.append(augment.AugParams(scale, flip, top, left, oh, ow))
        out = aug(imgs, params=params)
        torch.cuda.synchronize()
        assert out.shape == (len(rows), 3, crop, crop) and out.dtype == torch.float32
        for k, (i, row) in enumerate(rows):
            want = A.augment(imgs[k], float(row[3]), int(row[4]), int(row[5]), int(row[6]), crop)
            assert torch.equal(out[k].cpu(), torch.from_numpy(want)), i
            assert zlib.crc32(_u8(out[k].cpu().numpy()).tobytes()) == int(row[7]), i


@pytest.mark.gpu
def test_gpu_augment_random_draws_full_size(hific, dev):
    """Batch of 16 photo-sized images with the reference's random draws: output in [0,1], equals the oracle on a
    sample, normalised variant = (x - .5) / .5, and is deterministic for a fixed seed."""
    from hific_amd.helpers import augment
    rng = np.random.default_rng(1)
    imgs = [G.image(500 + i, int(rng.integers(300, 700)), int(rng.integers(300, 900))) for i in range(16)]
    aug = augment.GpuAugmenter(crop_size=256, device=dev)
    ps = [augment.draw_params(im.shape[0], im.shape[1], 256, np.random.RandomState(i), torch.Generator().manual_seed(i))
          for i, im in enumerate(imgs)]
    out = aug(imgs, params=ps)
    out2 = aug(imgs, params=ps)
    outn = augment.GpuAugmenter(crop_size=256, normalize=True, device=dev)(imgs, params=ps)
    torch.cuda.synchronize()
    assert torch.equal(out, out2) and float(out.min()) >= 0.0 and float(out.max()) <= 1.0
    assert torch.equal(outn, (out - 0.5) / 0.5)
    for k in (0, 7, 15):
        p = ps[k]
        want = A.augment(imgs[k], p.scale, p.flip, p.top, p.left, 256)
        assert torch.equal(out[k].cpu(), torch.from_numpy(want)), k

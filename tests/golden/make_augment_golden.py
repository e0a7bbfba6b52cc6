"""Golden vectors for the augmentation path, made with Pillow itself - the third-party library
whose `Image.resize(..., BILINEAR)` the reference's transforms call (datasets.py:206-216 via torchvision 0.7).
Run in the build container:  python tests/golden/make_augment_golden.py   ->  tests/golden/augment_golden.npz
Each case stores the seed of its synthetic uint8 image, the draws (scale, flip, top, left) and a checksum + a corner
patch of PIL's result for:  flip -> resize((ceil(sH), ceil(sW))) -> crop -> /255.
"""
import math
import os
import zlib

import numpy as np
import PIL
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
# (H, W, crop, scale, flip, top, left): ragged sizes, up- and down-scaling, identity axes, extreme crops
CASES = [
    (300, 400, 256, 0.90, 0, 7, 33),
    (300, 400, 256, 0.8533333333333334, 1, 0, 0),         # ceil(0.8533.. * 300) = 256: no vertical slack
    (341, 512, 256, 0.75, 1, 0, 128),                      # lowest scale: 256 x 384
    (256, 256, 256, 1.0, 0, 0, 0),                         # scale 1: Pillow resamples nothing
    (256, 700, 256, 1.0, 1, 0, 444),                       # identity in y, identity in x too (scale 1), wide crop
    (200, 260, 256, 1.28, 0, 0, 10),                       # small image: up-scaling (scale_low = 256/200)
    (97, 131, 64, 0.83, 1, 9, 30),                         # small crop
    (601, 403, 256, 0.9499, 1, 314, 126),                  # last rows/columns of the resized image
]


def image(seed, H, W):
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:H, 0:W]
    grad = ((yy * 3 + xx * 5) % 256).astype(np.uint8)[..., None]
    return np.where(rng.random((H, W, 1)) < 0.5, base, grad).astype(np.uint8)


def pil_augment(a, crop, scale, flip, top, left):
    im = Image.fromarray(a)
    if flip:
        im = im.transpose(Image.FLIP_LEFT_RIGHT)
    H, W = a.shape[:2]
    oh, ow = math.ceil(scale * H), math.ceil(scale * W)
    im = im.resize((ow, oh), Image.BILINEAR)
    im = im.crop((left, top, left + crop, top + crop))
    return np.asarray(im)


def main():
    out = {"pillow_version": np.array(PIL.__version__)}
    meta = []
    for i, (H, W, crop, scale, flip, top, left) in enumerate(CASES):
        r = pil_augment(image(100 + i, H, W), crop, scale, flip, top, left)
        assert r.shape == (crop, crop, 3), r.shape
        meta.append((H, W, crop, scale, flip, top, left, zlib.crc32(r.tobytes())))
        out[f"patch{i}"] = r[:8, :8].copy()
        out[f"last{i}"] = r[-4:, -4:].copy()
    out["cases"] = np.array(meta, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "augment_golden.npz"), **out)
    print("wrote", len(CASES), "cases with Pillow", PIL.__version__)


if __name__ == "__main__":
    main()

"""Convert the reference's LPIPS v0.1 linear-head weights (data file, not code:
/root/reference/src/loss/perceptual_similarity/weights/v0.1/{alex,vgg}.pth) into a plain .npz that ships with the
package, so PerceptualLoss works where /root/reference does not exist (the GPU box).  Run in the build container:
    python tools/extract_lpips_lin.py
"""
import os
import sys
import numpy as np
import torch

REF = os.environ.get("HIFIC_REFERENCE", "/root/reference")
src_dir = os.path.join(REF, "src/loss/perceptual_similarity/weights/v0.1")
dst_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..",
                       "high-fidelity-generative-compression_amd", "loss", "weights")
for net in ("alex", "vgg"):
    sd = torch.load(os.path.join(src_dir, net + ".pth"), map_location="cpu")
    out = {}
    for k, v in sd.items():          # lin{k}.model.1.weight : (1, C, 1, 1)
        out[k.split(".")[0]] = v.reshape(-1).numpy().astype(np.float32)
    path = os.path.join(dst_dir, f"lpips_{net}_lin_v0.1.npz")
    np.savez(path, **out)
    print(path, {k: a.shape for k, a in out.items()}, file=sys.stderr)

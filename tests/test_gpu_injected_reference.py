"""The headline drop-in claim, end to end on the device: `inject.patch_reference()` makes the reference's OWN
`src.model.Model` (src/model.py:68-105 builds the modules, :346-387 is the training forward) run on the HIP modules.

One float32 training step (G-turn of compression_gan: forward, both losses, backward) through the reference's stitcher
must equal the same step through this package's mirror (`hific_amd.Model`), and checkpoints must cross over in both
directions with the reference's state_dict layout.

The reference's Python sources reach the GPU box as `oracle/_ref/reference_src.tar.gz` (made by oracle/make_ref.py
from /root/reference; git-ignored test infrastructure) and are unpacked into a temporary directory here.
"""
import logging
import os
import sys
import tarfile

import pytest
import torch

from oracle import hific_oracle as O

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAR = os.path.join(ROOT, "oracle", "_ref", "reference_src.tar.gz")
N_RES = 9


def _relerr(a, b):
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-20)


@pytest.fixture(scope="module")
def ref_ns(tmp_path_factory):
    if os.path.isdir("/root/reference/src"):
        root = "/root/reference"
    elif os.path.exists(TAR):
        root = str(tmp_path_factory.mktemp("reference"))
        with tarfile.open(TAR) as t:
            t.extractall(root)
    else:
        pytest.skip("oracle/_ref/reference_src.tar.gz absent (run `python oracle/make_ref.py` where /root/reference exists)")
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ref_loader
    ns = ref_loader.load(root)
    import hific_amd.inject as inject
    patched = inject.patch_reference()
    assert "src.network.encoder.Encoder" in patched
    return ns


def _ref_args(ns, **over):
    cfg = ns.default_config
    d = {}
    for klass in reversed(cfg.hific_args.__mro__):
        d.update({k: v for k, v in vars(klass).items() if not k.startswith("__")})
    d.update(over)
    return ns.utils.Struct(**d)


def _step(model, x, noises):
    model.Hyperprior._draw_noise = lambda t: noises.pop(0)
    losses, inter = model(x, train_generator=True, return_intermediates=True, writeout=False)
    losses["compression"].backward()
    torch.cuda.synchronize()
    return losses, inter


def test_reference_model_on_hip_modules_equals_mirror(hific, dev, ref_ns, tmp_path):
    import hific_amd
    from hific_amd.default_config import make_args, hific_args, ModelTypes
    ns = ref_ns
    hific.set_compute_dtype(torch.float32)
    cfg = ns.default_config
    B, S = 2, 128
    dims = dict(batch_size=B, image_dims=(3, S, S), latent_dims=(220, S // 16, S // 16), n_residual_blocks=N_RES)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)          # the LPIPS seeded-backbone warning: expected here
        ref = ns.model.Model(_ref_args(ns, **dims), logging.getLogger("hific_ref"), model_mode=cfg.ModelModes.TRAINING,
                             model_type=cfg.ModelTypes.COMPRESSION_GAN)
    # the reference's constructor really built OUR modules
    from hific_amd.network.encoder import Encoder
    from hific_amd.hyperprior import Hyperprior
    assert isinstance(ref.Encoder, Encoder) and isinstance(ref.Hyperprior, Hyperprior)
    mir = hific_amd.Model(make_args(hific_args, **dims), model_type=ModelTypes.COMPRESSION_GAN,
                          allow_random_lpips_backbone=True)
    sd = O.make_state_dict(seed=0, gan=True, n_res=N_RES)
    ref.load_state_dict(sd, strict=True)
    mir.load_state_dict(sd, strict=True)
    bb = O.make_alex_backbone()
    ref.perceptual_loss.load_backbone_state_dict(bb)
    mir.perceptual_loss.load_backbone_state_dict(bb)
    ref, mir = ref.to(dev).train(), mir.to(dev).train()
    x = O.make_image(1, B, S, S).to(dev)
    nh, nl = O.make_noise(6, (B, 320, S // 64, S // 64)).to(dev), O.make_noise(7, (B, 220, S // 16, S // 16)).to(dev)
    lr, ir = _step(ref, x, [nh, nl])
    lm, im = _step(mir, x, [nh, nl])
    # same kernels, same order: the forward is identical; gradients may differ in the association of fan-in sums
    assert float(lr["compression"]) == float(lm["compression"]) and float(lr["disc"]) == float(lm["disc"])
    assert torch.equal(ir.reconstruction, im.reconstruction) and torch.equal(ir.latents_quantized, im.latents_quantized)
    pr, pm = dict(ref.named_parameters()), dict(mir.named_parameters())
    assert set(pr) == set(pm)
    worst = 0.0
    for k in pr:
        assert pr[k].grad is not None and pm[k].grad is not None, k
        worst = max(worst, _relerr(pr[k].grad, pm[k].grad))
    assert worst < 1e-5, worst
    assert torch.equal(ref.Discriminator.conv2.weight_u, mir.Discriminator.conv2.weight_u)
    # checkpoints in the reference's layout cross over in both directions (utils.save_model stores model.state_dict())
    path = str(tmp_path / "ckpt.pt")
    torch.save({"model_state_dict": ref.state_dict()}, path)
    mir.load_state_dict(torch.load(path)["model_state_dict"], strict=True)
    ref.load_state_dict(mir.state_dict(), strict=True)
    assert list(ref.state_dict().keys()) == list(mir.state_dict().keys()) and len(ref.state_dict()) == 168

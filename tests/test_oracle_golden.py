"""Pins the oracle (oracle/hific_oracle.py) to outputs of the REAL reference stored in tests/golden/
(made by tests/golden/make_golden.py from /root/reference).  Runs anywhere (CPU only, no reference needed)."""
import os

import pytest
import torch

from oracle import hific_oracle as O

GOLD = torch.load(os.path.join(os.path.dirname(__file__), "golden", "reference_outputs.pt"), weights_only=False)


def _lins():
    import numpy as np
    p = os.path.join(os.path.dirname(os.path.dirname(__file__)), "high-fidelity-generative-compression_amd", "loss",
                     "weights", "lpips_alex_lin_v0.1.npz")
    w = np.load(p)
    return [torch.from_numpy(w[f"lin{i}"].copy()) for i in range(5)]


def _close(a, b, rel=2e-4):
    return abs(a - b) <= rel * max(abs(b), 1e-12)


@pytest.mark.parametrize("case", ["compression_train", "compression_eval", "gan_train_G", "gan_train_D"])
def test_model_forward_backward_matches_reference(case):
    g = GOLD[case]
    s = g["seeds"]
    gan = g["gan"]
    sd = O.make_state_dict(seed=s["sd"], gan=gan)
    watch = list(g.get("grad_norms", {}))
    sdr = {k: (v.clone().requires_grad_(True) if k in watch else v) for k, v in sd.items()}
    x = O.make_image(s["image"], s["B"], s["H"], s["H"])
    nh, nl = O.make_noise(s["noise_h"], (2, 320, 2, 2)), O.make_noise(s["noise_l"], (2, 220, 8, 8))
    out = O.model_forward(sdr, O.make_alex_backbone(), _lins(), x, step_counter=1, training=g["training"], gan=gan,
                          train_generator=g["train_generator"], noise_hyper=nh, noise_latent=nl)
    assert _close(float(out["compression"]), g["compression"])
    assert _close(float(out["hyperinfo"].total_nbpp), g["n_bpp"])
    assert _close(float(out["hyperinfo"].total_qbpp), g["q_bpp"])
    rec = out["reconstruction"].detach()
    assert torch.allclose(rec[:, :, :6, :6], g["recon_patch"], rtol=1e-3, atol=1e-4)
    assert _close(float(rec.mean()), g["recon_mean"], 1e-3) and _close(float(rec.std()), g["recon_std"], 1e-3)
    dec = out["hyperinfo"].decoded.detach()
    assert torch.allclose(dec[:, :4, :3, :3], g["latents_patch"], rtol=1e-4, atol=1e-4)
    if g["training"]:
        (out["compression"] if g["train_generator"] else out["disc"]).backward()
        for k, n in g["grad_norms"].items():
            assert _close(float(sdr[k].grad.norm()), n, 2e-3), k
    if gan:
        assert _close(float(out["disc"]), g["disc"])
        if "weight_u_after" in g:
            assert torch.allclose(out["new_uv"]["Discriminator.conv3.weight_u"], g["weight_u_after"], atol=1e-5)


def test_primitives_match_reference():
    x = O.make_noise(11, (2, 12, 5, 7)) * 3
    y = O.channel_norm(x, O.make_noise(12, (1, 12, 1, 1)) + 1.2, O.make_noise(13, (1, 12, 1, 1)))
    assert torch.allclose(y, GOLD["channelnorm"]["y"], rtol=1e-5, atol=1e-6)
    t = (O.make_noise(14, (64,)) * 2).requires_grad_(True)
    z = O.lower_bound_toward(t, 0.11)
    z.backward(O.make_noise(15, (64,)))
    assert torch.equal(z.detach(), GOLD["lower_bound"]["y"]) and torch.equal(t.grad, GOLD["lower_bound"]["dx"])
    got = [float(O.get_scheduled_params(2.0, dict(vals=[2., 1.], steps=[50000]), s)) for s in (0, 1, 49999, 50000, 70000)]
    assert got == GOLD["sched"]


def test_quantized_indices_are_integers():
    y = O.make_noise(1, (2, 22, 4, 4)) * 9
    mu = O.make_noise(2, (2, 22, 4, 4))
    idx = O.quantized_indices(y, mu)
    assert idx.dtype == torch.int64
    assert torch.equal(idx.float(), torch.floor(y - mu + 0.5))

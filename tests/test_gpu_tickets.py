"""Tickets (include/hific_hip.h "tickets", csrc/common.h): the last-arriving workgroup of a partial-sum kernel runs the second
stage of the reduction inside the same launch.  With the stream's ticket buffer registered (lib.workspace does it) against the
two-launch forms (lib.set_tickets(False)): scalar loss sums, channel sums and LPIPS tap sums bit for bit (the ChannelNorm
parameter sums keep their separate launch: also compared here, the switch must not touch them); hundreds of back-to-back
launches leave the counters zero."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _restore(hific):
    from hific_amd import lib
    yield
    lib.set_tickets(True)


def _ab(fn):
    from hific_amd import lib
    lib.set_tickets(True)
    a = fn()
    lib.set_tickets(False)
    b = fn()
    lib.set_tickets(True)
    torch.cuda.synchronize()
    return a, b


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_scalar_and_channel_sums_are_bit_identical(hific, dev, dt):
    from hific_amd import ops, lib
    hific.set_compute_dtype(dt)
    g = torch.Generator(device=dev).manual_seed(1)
    a = torch.randn(4, 37, 50, 40, generator=g, device=dev).to(dt)
    b = torch.randn(4, 37, 50, 40, generator=g, device=dev)
    z = torch.randn(4096 + 13, generator=g, device=dev)
    p = torch.rand(3, 220, 16, 16, generator=g, device=dev) + 1e-3
    lib.workspace(dev)

    def run():
        out = [ops.MSEFn.apply(a, b, 255.0).detach().clone(),
               ops.BCELogitsFn.apply(z, 1.0).detach().clone(), ops.BCELogitsFn.apply(z, 0.0).detach().clone(),
               ops.LsqSigmoidFn.apply(z, 1.0).detach().clone(),
               ops.LogSumFn.apply(p, 1e-9, -0.37).detach().clone()]
        cs = torch.empty(37, device=dev)
        wsb = lib.workspace(dev)
        lib.call("hific_channel_sum", a.data_ptr(), cs.data_ptr(), 4, 37, 2000, 0, lib.dtype_code(a), wsb.data_ptr(),
                 wsb.numel(), lib.stream())
        out.append(cs.clone())
        lib.call("hific_channel_sum", a.data_ptr(), cs.data_ptr(), 4, 37, 2000, 1, lib.dtype_code(a), wsb.data_ptr(),
                 wsb.numel(), lib.stream())
        out.append(cs.clone())
        return out
    x, y = _ab(run)
    for u, v in zip(x, y):
        assert torch.equal(u, v)
    ref = (a.float() * 255 - b * 255).pow(2).mean()
    assert abs(float(x[0]) - float(ref)) < 1e-4 * float(ref)
    assert torch.allclose(x[5], a.float().sum(dim=(0, 2, 3)), rtol=1e-4, atol=1e-2)
    assert torch.allclose(x[6], 2 * a.float().sum(dim=(0, 2, 3)), rtol=1e-4, atol=2e-2)
    hific.set_compute_dtype(torch.float32)


@pytest.mark.parametrize("N,C,H,W,dt,relu", [(16, 960, 16, 16, torch.bfloat16, 1), (2, 60, 64, 64, torch.bfloat16, 1),
                                             (2, 220, 8, 8, torch.float32, 0), (3, 480, 32, 32, torch.bfloat16, 0),
                                             (1, 37, 17, 5, torch.float32, 1)])
def test_channelnorm_parameter_gradients(hific, dev, N, C, H, W, dt, relu):
    from hific_amd import ops
    hific.set_compute_dtype(dt)
    g = torch.Generator(device=dev).manual_seed(2)
    x = torch.randn(N, C, H, W, generator=g, device=dev).to(dt)
    dy = torch.randn(N, C, H, W, generator=g, device=dev).to(dt)
    gamma = (torch.rand(1, C, 1, 1, generator=g, device=dev) + 0.5).requires_grad_(True)
    beta = (torch.randn(1, C, 1, 1, generator=g, device=dev) * 0.2).requires_grad_(True)
    bias = torch.zeros(C, device=dev, requires_grad=True)

    def run():
        gamma.grad = beta.grad = bias.grad = None
        xx = x.clone().requires_grad_(True)
        y = ops.channel_norm(xx, gamma, beta, 1e-3, relu=bool(relu), prev_bias=bias)
        y.backward(dy)
        return [xx.grad.clone(), gamma.grad.clone(), beta.grad.clone(), bias.grad.clone()]
    (dx1, dg1, db1, dp1), (dx2, dg2, db2, dp2) = _ab(run)
    assert torch.equal(dx1, dx2)
    for u, v in ((dg1, dg2), (db1, db2), (dp1, dp2)):
        scale = float(v.abs().max()) + 1e-20
        assert float((u - v).abs().max()) <= 2e-6 * scale + 1e-6 * float(v.abs().mean()) * (N * H * W) ** 0.5, float((u - v).abs().max()) / scale
    again = run()
    assert all(torch.equal(p, q) for p, q in zip((dx1, dg1, db1, dp1), again))          # fixed summation order
    hific.set_compute_dtype(torch.float32)


def test_lpips_tap_sums_bit_identical_and_counters_self_clean(hific, dev):
    from hific_amd import lib
    g = torch.Generator(device=dev).manual_seed(3)
    B, C, HW = 5, 192, 27 * 27
    f = torch.randn(2 * B, C, HW, generator=g, device=dev)
    w = torch.rand(C, generator=g, device=dev)
    wsb = lib.workspace(dev)

    def run():
        val = torch.zeros(B, device=dev)
        for acc in (0, 1):
            lib.call("hific_lpips_tap_fwd", f.data_ptr(), w.data_ptr(), val.data_ptr(), B, C, HW, acc, lib.HIFIC_F32,
                     wsb.data_ptr(), wsb.numel(), lib.stream())
        return val.clone()
    a, b = _ab(run)
    assert torch.equal(a, b)
    # 300 launches back to back: every one must find its counters at zero
    lib.set_tickets(True)
    outs = [run() for _ in range(150)]
    torch.cuda.synchronize()
    assert all(torch.equal(o, a) for o in outs)
    key = next(k for k in lib._tickets)
    assert int(lib._tickets[key].view(torch.int32).abs().sum()) == 0


@pytest.mark.parametrize("shape,factor", [((2, 3, 37, 50), 16), ((1, 220, 7, 9), 4), ((2, 3, 64, 64), 16)])
def test_pad_factor_on_the_device_equals_reflection_pad(hific, dev, shape, factor):
    """helpers.utils.pad_factor (src/helpers/utils.py:50-62, the EVALUATION path's pad to a multiple of 16 / 4) through
    hific_pad2d: bit-identical to F.pad(mode='reflect'), no-op when already aligned."""
    import torch.nn.functional as F
    from hific_amd.helpers import utils
    x = torch.randn(*shape, device=dev)
    y = utils.pad_factor(x, x.shape[2:], factor)
    H, W = shape[2], shape[3]
    ph, pw = (factor - H % factor) % factor, (factor - W % factor) % factor
    ref = F.pad(x, (0, pw, 0, ph), mode="reflect") if (ph or pw) else x
    assert y.shape == ref.shape and torch.equal(y, ref)


@pytest.mark.parametrize("target_rate,gan", [(None, True), (1e3, True), (None, False)], ids=["gan_lambda_A", "gan_lambda_B", "no_gan"])
def test_fused_loss_composition_equals_the_torch_glue(hific, dev, monkeypatch, target_rate, gan):
    """ops.LossCombineFn (hific_loss_combine_fwd / _bwd: k_M mse + k_P mean(lpips) + lambda(q_bpp) n_bpp [+ beta G_loss] with the
    rate rule of src/loss/losses.py:8-28 on the device) against the reference-shaped composition on zero-dimensional torch
    tensors (Model.compression_loss + `loss + beta * G_loss`): same loss to float32 rounding, same gradients for every
    parameter - both branches of the rate rule (q_bpp above / below the target) and the model without a Discriminator."""
    import hific_amd
    from hific_amd import model as model_mod
    from hific_amd.default_config import make_args, hific_args, mse_lpips_args, ModelTypes
    from oracle import hific_oracle as O
    hific.set_compute_dtype(torch.float32)
    kw = dict(batch_size=2, image_dims=(3, 128, 128), latent_dims=(220, 8, 8), n_residual_blocks=2)
    if target_rate is not None:
        kw["target_rate"] = target_rate
    args = make_args(hific_args if gan else mse_lpips_args, **kw)
    x = O.make_image(1, 2, 128, 128).to(dev)
    nh, nl = O.make_noise(6, (2, 320, 2, 2)).to(dev), O.make_noise(7, (2, 220, 8, 8)).to(dev)
    res = []
    for fused in (True, False):
        monkeypatch.setattr(model_mod, "_FUSED_LOSS", fused)
        torch.manual_seed(0)
        m = hific_amd.Model(args, model_type=ModelTypes.COMPRESSION_GAN if gan else ModelTypes.COMPRESSION,
                            device_rate_select=True, allow_random_lpips_backbone=True)
        m.load_state_dict(O.make_state_dict(seed=0, gan=gan, n_res=2), strict=True)
        m.perceptual_loss.load_backbone_state_dict(O.make_alex_backbone())
        m = m.to(dev).train()
        noises = [nh, nl]
        m.Hyperprior._draw_noise = lambda t: noises.pop(0)
        losses = m(x, train_generator=True, writeout=False)
        assert m._fused_loss_ok(x) == fused
        losses["compression"].backward()
        torch.cuda.synchronize()
        res.append((float(losses["compression"].detach()), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}))
    (la, ga), (lb, gb) = res
    assert abs(la - lb) <= 2e-6 * abs(lb), (la, lb)
    assert set(ga) == set(gb)
    for k in ga:
        scale = float(gb[k].abs().max()) + 1e-30
        assert float((ga[k] - gb[k]).abs().max()) <= 2e-5 * scale, (k, float((ga[k] - gb[k]).abs().max()) / scale)


def test_spectral_norm_backward_uses_the_uv_of_its_own_forward(hific, dev):
    """The batched power iteration writes copies of the post-iteration (u, v) behind [sigma, 1/sigma]
    (hific_spectral_norm_fwd_batch do_iter bit 1) and SNConv2dFn / D1StageFn save THOSE: a second training forward (which
    iterates the buffers in place) between a forward and its backward must not change that backward - as with torch's
    spectral_norm, which clones u and v (src/network/discriminator.py:46-62)."""
    from hific_amd.network.discriminator import Discriminator
    from oracle import hific_oracle as O
    hific.set_compute_dtype(torch.float32)
    sd = O.make_state_dict(seed=0, gan=True, n_res=2)
    real = O.make_image(9, 2, 64, 64).to(dev)
    gen = O.make_image(12, 2, 64, 64).to(dev)
    lat = (O.make_noise(10, (2, 220, 4, 4)) * 4).to(dev)
    grads = []
    for twice in (False, True):
        D = Discriminator((3, 64, 64), (220, 4, 4), C=220)
        D.load_state_dict({k[len("Discriminator."):]: v for k, v in sd.items() if k.startswith("Discriminator.")}, strict=True)
        D = D.to(dev).train()
        g = gen.clone().requires_grad_(True)
        _, logits = D.forward_pairs(real, g, lat)
        if twice:
            with torch.no_grad():
                D.forward_pairs(real, gen, lat)            # iterates weight_u / weight_v in place once more
        logits.sum().backward()
        torch.cuda.synchronize()
        grads.append([g.grad.clone()] + [p.grad.clone() for p in D.parameters()])
    for a, b in zip(*grads):
        assert torch.equal(a, b)

"""GPU parity of the HBM-bound kernels (norm / entropy model / LPIPS taps / pooling / losses / Adam / spectral norm)
against torch CPU float32 restatements of the reference ops (oracle/hific_oracle.py primitives)."""
import math
import os

import pytest
import torch
import torch.nn.functional as F

from oracle import hific_oracle as O

pytestmark = pytest.mark.gpu


def _rnd(shape, seed, lo=-1.0, hi=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(shape, generator=g) * (hi - lo) + lo


def _relerr(a, b):
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-20)


@pytest.mark.parametrize("dt,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)], ids=["f32", "bf16"])
@pytest.mark.parametrize("shape,relu", [((2, 60, 24, 24), True), ((2, 960, 16, 16), False), ((3, 220, 5, 7), True),
                                        ((4, 8, 250, 256), True), ((1, 1100, 3, 5), False),
                                        # many channels on narrow planes (residual blocks; 640: three channels per thread)
                                        ((3, 960, 16, 16), True), ((2, 640, 8, 8), True)])
def test_channelnorm(hific, dev, shape, relu, dt, tol):
    from hific_amd import ops
    x = _rnd(shape, 1, -2, 2)
    if dt == torch.bfloat16:
        x = x.to(dt).float()
    C = shape[1]
    gamma = _rnd((1, C, 1, 1), 2, 0.5, 1.5)
    beta = _rnd((1, C, 1, 1), 3, -0.3, 0.3)
    xr, gr, br = x.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    yr = O.channel_norm(xr, gr, br)
    if relu:
        yr = F.relu(yr)
    gy = _rnd(shape, 4)
    if dt == torch.bfloat16:
        gy = gy.to(dt).float()
    yr.backward(gy)
    xd = x.to(dev).to(dt).requires_grad_(True)
    gd, bd = gamma.to(dev).requires_grad_(True), beta.to(dev).requires_grad_(True)
    y = ops.channel_norm(xd, gd, bd, 1e-3, relu=relu)
    y.backward(gy.to(dev).to(dt))
    torch.cuda.synchronize()
    assert _relerr(y.detach().float().cpu(), yr.detach()) < tol
    assert _relerr(xd.grad.float().cpu(), xr.grad) < tol * 3
    assert _relerr(gd.grad.cpu(), gr.grad) < tol * 3
    assert _relerr(bd.grad.cpu(), br.grad) < tol * 3


@pytest.mark.parametrize("shape", [(3, 960, 16, 16), (2, 640, 8, 8), (2, 60, 24, 24)])
def test_channelnorm_backward_returns_producer_bias_gradient(hific, dev, shape):
    """ChannelNorm backward with `prev_bias` (normalisation.channel.fuse_bias_grad): the third output is sum_{n,h,w} dx as
    the tensor holds it (bf16-rounded), i.e. the bias gradient of the convolution whose output the norm consumes."""
    from hific_amd import ops
    dt = torch.bfloat16
    N, C, H, W = shape
    x = _rnd(shape, 11, -2, 2).to(dev).to(dt).requires_grad_(True)
    gamma = _rnd((1, C, 1, 1), 12, 0.5, 1.5).to(dev).requires_grad_(True)
    beta = _rnd((1, C, 1, 1), 13, -0.3, 0.3).to(dev).requires_grad_(True)
    pbias = torch.zeros(C, device=dev, requires_grad=True)
    gy = _rnd(shape, 14).to(dev).to(dt)
    y = ops.channel_norm(x, gamma, beta, 1e-3, relu=True, prev_bias=pbias)
    y.backward(gy)
    x2 = x.detach().clone().requires_grad_(True)
    g2, b2 = gamma.detach().clone().requires_grad_(True), beta.detach().clone().requires_grad_(True)
    y2 = ops.channel_norm(x2, g2, b2, 1e-3, relu=True)
    y2.backward(gy)
    torch.cuda.synchronize()
    assert torch.equal(x.grad, x2.grad) and torch.equal(gamma.grad, g2.grad) and torch.equal(beta.grad, b2.grad)
    want = x.grad.float().sum(dim=(0, 2, 3))
    scale = float(x.grad.float().abs().sum(dim=(0, 2, 3)).max())
    assert float((pbias.grad - want).abs().max()) <= 1e-5 * scale


def test_factorized_likelihood(hific, dev):
    from hific_amd import ops
    N, C, H, W = 3, 20, 4, 4
    sd = {k: v for k, v in O.make_state_dict(seed=5, C=12, N=C, n_res=0, gan=False).items()
          if "hyperlatent_likelihood" in k}
    pref = "Hyperprior.hyperlatent_likelihood."
    z = _rnd((N, C, H, W), 7, -4, 4)
    z[0, 0, 0, 0] = 40.0       # deep tail -> lower-bounded likelihood, exercises the LowerBoundToward gate
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    zr = z.clone().requires_grad_(True)
    lik_r = O.factorized_likelihood(sdr, zr, pref)
    gl = _rnd((N, C, H, W), 8)
    lik_r.backward(gl)
    params = [sd[pref + f"{n}_{k}"].to(dev).requires_grad_(True) for n in ("H", "a", "b") for k in range(4)]
    zd = z.to(dev).requires_grad_(True)
    lik = ops.FactorizedLikFn.apply(zd, 1e-9, *params)
    lik.backward(gl.to(dev))
    torch.cuda.synchronize()
    assert _relerr(lik.detach().cpu(), lik_r.detach()) < 1e-4
    assert _relerr(zd.grad.cpu(), zr.grad) < 1e-3
    i = 0
    for n in ("H", "a", "b"):
        for k in range(4):
            assert _relerr(params[i].grad.cpu(), sdr[pref + f"{n}_{k}"].grad) < 2e-3, (n, k)
            i += 1


@pytest.mark.parametrize("ltype", ["gaussian", "logistic"])
def test_latent_likelihood_and_entropy(hific, dev, ltype):
    from hific_amd import ops
    shape = (2, 22, 8, 8)
    x = _rnd(shape, 1, -6, 6)
    mean = _rnd(shape, 2, -2, 2)
    raw = _rnd(shape, 3, -0.5, 2.0)       # part below the 0.11 bound
    xr, mr, rr = (t.clone().requires_grad_(True) for t in (x, mean, raw))
    sc_r = O.lower_bound_toward(rr, 0.11)
    lik_r = O.latent_likelihood(xr, mr, sc_r, ltype)
    _, bpp_r = O.estimate_entropy(lik_r, (128, 128))
    bpp_r.backward()
    xd, md, rd = (t.to(dev).requires_grad_(True) for t in (x, mean, raw))
    sc = ops.LowerBoundFn.apply(rd, 0.11)
    lik = ops.GaussLikFn.apply(xd, md, sc, 1e-9, 1 if ltype == "logistic" else 0)
    nbits = ops.LogSumFn.apply(lik, 1e-9, 1.0 / (shape[0] * -math.log(2.0)))
    bpp = nbits / (128 * 128)
    bpp.backward()
    torch.cuda.synchronize()
    assert _relerr(lik.detach().cpu(), lik_r.detach()) < 1e-4
    assert abs(float(bpp) - float(bpp_r)) < 1e-4 * abs(float(bpp_r))
    assert _relerr(xd.grad.cpu(), xr.grad) < 2e-3
    assert _relerr(md.grad.cpu(), mr.grad) < 2e-3
    assert _relerr(rd.grad.cpu(), rr.grad) < 2e-3


def test_round_ops_bit_exact(hific, dev):
    from hific_amd import ops
    x = _rnd((2, 22, 8, 8), 1, -8, 8)
    m = _rnd((2, 22, 8, 8), 2, -1, 1)
    q = ops.RoundFn.apply(x.to(dev), m.to(dev)).cpu()
    assert torch.equal(q, torch.floor(x - m + 0.5) + m)
    q0 = ops.RoundFn.apply(x.to(dev), None).cpu()
    assert torch.equal(q0, torch.floor(x + 0.5))
    xd = x.to(dev).requires_grad_(True)
    md = m.to(dev).requires_grad_(True)
    st = ops.RoundSTFn.apply(xd, md)
    st.sum().backward()
    assert torch.equal(st.detach().cpu(), torch.floor(x - m + 0.5) + m)
    assert torch.equal(xd.grad.cpu(), torch.ones_like(x)) and md.grad is None


@pytest.mark.parametrize("dt,tol", [(torch.float32, 1e-4), (torch.bfloat16, 3e-2)], ids=["f32", "bf16"])
def test_lpips_forward_backward(hific, dev, dt, tol):
    from hific_amd.loss.perceptual_loss import PerceptualLoss
    hific.set_compute_dtype(dt)
    B, H = 2, 96
    bb = O.make_alex_backbone()
    pl = PerceptualLoss().to(dev)
    pl.load_backbone_state_dict(bb)
    lins = [getattr(pl, f"lin{i}").cpu() for i in range(5)]
    target = O.make_image(3, B, H, H)
    pred = (target + 0.1 * _rnd((B, 3, H, H), 5)).clamp(0, 1)
    pr = pred.clone().requires_grad_(True)
    vr = O.lpips_forward(bb, lins, pr, target, normalize=True)
    vr.mean().backward()
    pd = pred.to(dev).to(dt).requires_grad_(True)
    v = pl(pd, target.to(dev), normalize=True)
    v.mean().backward()
    torch.cuda.synchronize()
    assert v.shape == (B, 1, 1, 1)
    assert _relerr(v.detach().cpu(), vr.detach()) < tol
    assert _relerr(pd.grad.float().cpu(), pr.grad) < max(tol * 5, 1e-3)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_lpips_target_prefetch_is_bit_identical(hific, dev, dt):
    """PerceptualLoss.prefetch_target computes the target half of the feature maps ahead of time (model.py starts it on
    a second stream): value and gradient must equal the single 2B-batch pass bit for bit; a forward with another target
    must ignore the stale prefetch."""
    from hific_amd.loss.perceptual_loss import PerceptualLoss
    hific.set_compute_dtype(dt)
    B, H = 4, 96
    pl = PerceptualLoss(allow_random_backbone=True).to(dev)
    target = O.make_image(3, B, H, H).to(dev)
    other = O.make_image(4, B, H, H).to(dev)
    pred = (target + 0.1 * _rnd((B, 3, H, H), 5).to(dev)).clamp(0, 1).to(dt)

    def run(prefetch, tgt):
        pd = pred.clone().requires_grad_(True)
        if prefetch is not None:
            s2 = torch.cuda.Stream(device=dev)
            s2.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(s2):
                pl.prefetch_target(prefetch, normalize=True)
        v = pl(pd, tgt, normalize=True)
        v.mean().backward()
        torch.cuda.synchronize()
        return v.detach().clone(), pd.grad.clone()

    v0, g0 = run(None, target)
    v1, g1 = run(target, target)
    v2, g2 = run(other, target)              # stale prefetch (different tensor): ignored
    assert torch.equal(v0, v1) and torch.equal(g0, g1)
    assert torch.equal(v0, v2) and torch.equal(g0, g2)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_maxpool_fwd_bwd(hific, dev, dt):
    from hific_amd import lib
    x = F.relu(_rnd((2, 5, 15, 31), 1))
    x = x.to(dt).float()
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool2d(xr, 3, 2)
    gy = _rnd(tuple(yr.shape), 2).to(dt).float()
    yr.backward(gy)
    xd = x.to(dev).to(dt)
    y = torch.empty(yr.shape, dtype=dt, device=dev)
    lib.call("hific_maxpool3s2_fwd", xd.data_ptr(), y.data_ptr(), 10, 15, 31, lib.dtype_code(xd), lib.stream())
    dx = torch.empty_like(xd)
    gyd = gy.to(dev).to(dt)
    lib.call("hific_maxpool3s2_bwd", xd.data_ptr(), gyd.data_ptr(), dx.data_ptr(), 10, 15, 31, lib.dtype_code(xd),
             lib.stream())
    torch.cuda.synchronize()
    assert torch.equal(y.float().cpu(), yr.detach())
    assert _relerr(dx.float().cpu(), xr.grad) < (1e-6 if dt == torch.float32 else 1e-2)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("hw", [(16, 32), (15, 31)], ids=["even", "odd"])
def test_maxpool2s2_fwd_bwd(hific, dev, dt, hw):
    """nn.MaxPool2d(2, 2) of VGG16.features: values exact, gradient to the first maximum of a window (ties included:
    post-ReLU maps are full of equal zeros), nothing to the uncovered last row / column of odd planes."""
    from hific_amd import lib
    H, W = hw
    x = F.relu(_rnd((2, 5, H, W), 1))
    x = x.to(dt).float()
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool2d(xr, 2, 2)
    gy = _rnd(tuple(yr.shape), 2).to(dt).float()
    yr.backward(gy)
    xd = x.to(dev).to(dt)
    y = torch.empty(yr.shape, dtype=dt, device=dev)
    lib.call("hific_maxpool2s2_fwd", xd.data_ptr(), y.data_ptr(), 10, H, W, lib.dtype_code(xd), lib.stream())
    dx = torch.empty_like(xd)
    gyd = gy.to(dev).to(dt)
    lib.call("hific_maxpool2s2_bwd", xd.data_ptr(), gyd.data_ptr(), dx.data_ptr(), 10, H, W, lib.dtype_code(xd),
             lib.stream())
    torch.cuda.synchronize()
    assert torch.equal(y.float().cpu(), yr.detach())
    assert torch.equal(dx.float().cpu(), xr.grad)


@pytest.mark.parametrize("dt,tol", [(torch.float32, 1e-4), (torch.bfloat16, 5e-2)], ids=["f32", "bf16"])
def test_lpips_vgg_forward_backward(hific, dev, dt, tol):
    """PerceptualLoss(net='vgg') (networks_basic.py:36-38): value and gradient vs the oracle (which is pinned to the
    reference's VGG variant in tests/test_oracle_vs_reference.py::test_lpips_vgg_variant)."""
    from hific_amd.loss.perceptual_loss import PerceptualLoss
    hific.set_compute_dtype(dt)
    B, H = 2, 64
    bb = O.make_vgg_backbone()
    pl = PerceptualLoss(net='vgg', allow_random_backbone=True).to(dev)
    pl.load_backbone_state_dict(bb)
    lins = [getattr(pl, f"lin{i}").cpu() for i in range(5)]
    target = O.make_image(3, B, H, H)
    pred = (target + 0.1 * _rnd((B, 3, H, H), 5)).clamp(0, 1)
    pr = pred.clone().requires_grad_(True)
    vr = O.lpips_forward(bb, lins, pr, target, normalize=True, net="vgg")
    vr.mean().backward()
    pd = pred.to(dev).to(dt).requires_grad_(True)
    v = pl(pd, target.to(dev), normalize=True)
    v.mean().backward()
    torch.cuda.synchronize()
    assert v.shape == (B, 1, 1, 1)
    assert _relerr(v.detach().cpu(), vr.detach()) < tol
    assert _relerr(pd.grad.float().cpu(), pr.grad) < max(tol * 5, 1e-3)


def test_mse_and_bce(hific, dev):
    from hific_amd import ops
    a = _rnd((2, 3, 32, 32), 1, 0, 1)
    b = _rnd((2, 3, 32, 32), 2, 0, 1)
    ar = a.clone().requires_grad_(True)
    lr = torch.mean((ar * 255. - b * 255.) ** 2)
    lr.backward()
    ad = a.to(dev).requires_grad_(True)
    l = ops.MSEFn.apply(ad, b.to(dev), 255.0)
    l.backward()
    assert abs(float(l) - float(lr)) < 1e-5 * float(lr)
    assert _relerr(ad.grad.cpu(), ar.grad) < 1e-5
    z = _rnd((512,), 3, -5, 5)
    for target in (0.0, 1.0):
        zr = z.clone().requires_grad_(True)
        lr = F.binary_cross_entropy_with_logits(zr, torch.full_like(zr, target))
        lr.backward()
        zd = z.to(dev).requires_grad_(True)
        l = ops.BCELogitsFn.apply(zd, target)
        l.backward()
        assert abs(float(l) - float(lr)) < 1e-5 * abs(float(lr))
        assert _relerr(zd.grad.cpu(), zr.grad) < 1e-5


def test_adam_matches_torch(hific, dev):
    from hific_amd import ops
    p0 = _rnd((1000,), 1)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=1e-4)
    p = p0.to(dev)
    m = torch.zeros_like(p)
    v = torch.zeros_like(p)
    for step in range(1, 4):
        g = _rnd((1000,), 10 + step)
        ref.grad = g.clone()
        opt.step()
        ops.adam_step(p, g.to(dev), m, v, 1e-4, 0.9, 0.999, 1e-8, step)
    torch.cuda.synchronize()
    assert (p.cpu() - ref.detach()).abs().max().item() < 1e-7


def test_fused_adam_survives_module_moves_and_checkpoints(hific, dev):
    """FusedAdam over a ParamArena == torch.optim.Adam through: a model.cpu() -> .to(device) round trip between steps
    (the reference's save_model, utils.py:116-145), a parameter that receives no gradient in a step (zero gradient,
    not the previous step's), and a state_dict hand-over torch -> FusedAdam in the middle of training."""
    from hific_amd import optim
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(33, 17), torch.nn.Linear(17, 5)).to(dev)
    ref = torch.nn.Sequential(torch.nn.Linear(33, 17), torch.nn.Linear(17, 5))
    ref.load_state_dict({k: v.cpu() for k, v in net.state_dict().items()})
    ropt = torch.optim.Adam(ref.parameters(), lr=1e-3)
    opt = optim.FusedAdam(list(net.parameters()), lr=1e-3)

    def both_step(it, skip_last=False):
        for (pn, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
            g = _rnd(tuple(q.shape), 100 * it + q.numel())
            if skip_last and pn.startswith("1."):
                q.grad = torch.zeros_like(q)          # torch 1.6 zero_grad(): zero gradient, moments still decay
                continue
            q.grad = g.clone()
            s = p._hific_slot
            assert s.take() == 0
            s.grad.copy_(g.to(dev))
        ropt.step(); opt.step()
        ropt.zero_grad(); opt.zero_grad()

    both_step(1)
    both_step(2, skip_last=True)
    net.cpu(); net.to(dev)                           # parameters leave the arena ...
    both_step(3)                                     # ... and step() re-binds them instead of updating frozen copies
    for p, q in zip(net.parameters(), ref.parameters()):
        assert (p.detach().cpu() - q.detach()).abs().max().item() < 2e-7
    # hand-over: a fresh FusedAdam resumes from torch's own optimizer state_dict
    net2 = torch.nn.Sequential(torch.nn.Linear(33, 17), torch.nn.Linear(17, 5)).to(dev)
    net2.load_state_dict(ref.state_dict())
    opt2 = optim.FusedAdam(list(net2.parameters()), lr=9.0)
    opt2.load_state_dict(ropt.state_dict())
    net, opt = net2, opt2
    both_step(4)
    for p, q in zip(net2.parameters(), ref.parameters()):
        assert (p.detach().cpu() - q.detach()).abs().max().item() < 2e-7
    sd = opt2.state_dict()
    assert float(sd["state"][0]["step"]) == 4.0 and sd["param_groups"][0]["lr"] == 1e-3


def test_least_squares_gan_loss(hific, dev):
    """losses.py:43-50 on D = sigmoid(logits): values and the gradient w.r.t. the logits."""
    from collections import namedtuple
    from hific_amd.loss import losses
    D = namedtuple("D", ["D_real", "D_gen", "D_real_logits", "D_gen_logits"])
    zr, zg = _rnd((512,), 1) * 3, _rnd((512,), 2) * 3
    zr_r, zg_r = zr.clone().requires_grad_(True), zg.clone().requires_grad_(True)
    dr, dg = torch.sigmoid(zr_r), torch.sigmoid(zg_r)
    D_ref = 0.5 * (torch.mean((dr - 1.0) ** 2) + torch.mean(dg ** 2))
    G_ref = 0.5 * torch.mean((dg - 1.0) ** 2)
    (D_ref + 2.0 * G_ref).backward()
    zr_d, zg_d = zr.to(dev).requires_grad_(True), zg.to(dev).requires_grad_(True)
    D_loss, G_loss = losses.gan_losses("least_squares", D(None, None, zr_d, zg_d))
    (D_loss + 2.0 * G_loss).backward()
    torch.cuda.synchronize()
    assert abs(float(D_loss) - float(D_ref)) < 1e-6 and abs(float(G_loss) - float(G_ref)) < 1e-6
    assert _relerr(zr_d.grad.cpu(), zr_r.grad) < 1e-5 and _relerr(zg_d.grad.cpu(), zg_r.grad) < 1e-5
    with pytest.raises(ValueError):
        losses.gan_losses("hinge", D(None, None, zr_d, zg_d))


def test_spectral_norm_and_upcat(hific, dev):
    from hific_amd import ops
    w = _rnd((8, 5, 4, 4), 1)
    u = F.normalize(_rnd((8,), 2), dim=0)
    v = F.normalize(_rnd((80,), 3), dim=0)
    wn_r, u_r, v_r = O.spectral_norm_weight(w, u, v, training=True)
    ud, vd = u.to(dev).clone(), v.to(dev).clone()
    sig = ops.spectral_norm_power_iteration(w.to(dev), ud, vd, do_iter=True)
    torch.cuda.synchronize()
    assert _relerr(ud.cpu(), u_r) < 1e-5 and _relerr(vd.cpu(), v_r) < 1e-5
    assert _relerr(w / sig[0].cpu(), wn_r) < 1e-5
    img = _rnd((2, 3, 32, 32), 4)
    ctx = _rnd((2, 4, 2, 2), 5)
    ir, cr = img.clone().requires_grad_(True), ctx.clone().requires_grad_(True)
    outr = torch.cat((ir, F.interpolate(cr, scale_factor=16, mode="nearest")), dim=1)
    g = _rnd(tuple(outr.shape), 6)
    outr.backward(g)
    idv, cdv = img.to(dev).requires_grad_(True), ctx.to(dev).requires_grad_(True)
    out = ops.UpsampleConcatFn.apply(idv, cdv, 16)
    out.backward(g.to(dev))
    torch.cuda.synchronize()
    assert torch.equal(out.detach().cpu(), outr.detach())
    assert _relerr(idv.grad.cpu(), ir.grad) < 1e-6 and _relerr(cdv.grad.cpu(), cr.grad) < 1e-5


def test_compress_symbols_and_indices_bit_exact(hific, dev):
    """EVALUATION path, device half of `compress`: int32 symbols and table indices equal the
    oracle's (= the reference's, tests/test_oracle_vs_reference.py::test_symbols_and_indices) exactly, including
    rounding ties and scales exactly on table entries."""
    from hific_amd import ops
    for seed, shape in ((11, (2, 6, 5, 7)), (12, (1, 220, 16, 16)), (13, (3, 1, 1, 1))):
        lat, means, scales, table = O.make_symbol_inputs(seed, shape)
        sym, idx = ops.prior_symbols_and_indices(lat.to(dev), means.to(dev), scales.to(dev), table)
        torch.cuda.synchronize()
        assert sym.dtype == torch.int32 and idx.dtype == torch.int32
        assert torch.equal(sym.cpu(), O.prior_symbols(lat, means))
        assert torch.equal(idx.cpu(), O.prior_compute_indices(scales, table))
    z = torch.randn(2, 320, 4, 4) * 6
    z.view(-1)[::5] = torch.randint(-8, 8, (z.view(-1)[::5].numel(),)).float() + 0.5
    sym, idx = ops.hyper_symbols_and_indices(z.to(dev))
    torch.cuda.synchronize()
    s_o, i_o = O.hyper_symbols_and_indices(z)
    assert torch.equal(sym.cpu(), s_o) and torch.equal(idx.cpu(), i_o)


@pytest.mark.parametrize("shape", [(1, 24, 6, 5), (3, 8, 4, 4)])
def test_vectorised_coder_takes_device_tensors(hific, dev, shape):
    """compression.rans lays device tensors out on the device ((N,C,H,W) -> [steps][lanes]) and, with `device=...`, undoes it
    after the upload: bitstream and decoded symbols equal the all-host path's, for the batch-1 (steps = pixels) and the
    batch > 1 (steps = images) layouts."""
    import numpy as np
    from hific_amd.compression import rans
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tables_golden.npz"))
    cdf, cl, co = g["prior_CDF"].astype(np.uint32), g["prior_CDF_length"].astype(np.int32), g["prior_CDF_offset"].astype(np.int32)
    gen = torch.Generator().manual_seed(3)
    idx = torch.randint(0, cdf.shape[0], shape, generator=gen, dtype=torch.int32)
    sym = torch.round(torch.randn(shape, generator=gen) * 2).to(torch.int32)
    enc_h, cs_h = rans.ans_compress(sym.numpy(), idx.numpy(), cdf, cl, co, shape[1:], 16, vectorize=True)
    enc_d, cs_d = rans.ans_compress(sym.to(dev), idx.to(dev), cdf, cl, co, shape[1:], 16, vectorize=True)
    assert tuple(cs_h) == tuple(cs_d) and np.array_equal(enc_h, enc_d)
    dec_h = rans.ans_decompress(enc_h, idx.numpy(), cdf, cl, co, cs_h, 16, vectorize=True)
    dec_d = rans.ans_decompress(enc_d, idx.to(dev), cdf, cl, co, cs_d, 16, vectorize=True, device=dev)
    assert isinstance(dec_d, torch.Tensor) and dec_d.is_cuda and dec_d.dtype == torch.int32
    assert tuple(dec_d.shape) == shape and np.array_equal(dec_d.cpu().numpy(), dec_h)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_channelnorm_with_residual_equals_norm_then_add(hific, dev, dt):
    """hific_channelnorm_fwd_res (round 4: the ResidualBlock's add folded into its second norm, generator.py:44): bit-identical
    to the norm followed by hific_add, and the residual's gradient is the incoming gradient itself."""
    from hific_amd import ops
    x = _rnd((2, 960, 16, 16), 1).to(dev).to(dt).requires_grad_(True)
    r = _rnd((2, 960, 16, 16), 2).to(dev).to(dt).requires_grad_(True)
    g = (torch.rand(1, 960, 1, 1, device=dev) + 0.5).requires_grad_(True)
    b = (torch.rand(1, 960, 1, 1, device=dev) - 0.5).requires_grad_(True)
    y1 = ops.channel_norm(x, g, b, 1e-3, relu=False, resid=r)
    y2 = ops.add(ops.channel_norm(x, g, b, 1e-3, relu=False), r)
    assert torch.equal(y1, y2)
    dy = _rnd((2, 960, 16, 16), 3).to(dev).to(dt)
    gx1, gr1, gg1 = torch.autograd.grad(y1, (x, r, g), dy)
    gx2, gr2, gg2 = torch.autograd.grad(y2, (x, r, g), dy)
    torch.cuda.synchronize()
    assert torch.equal(gx1, gx2) and torch.equal(gr1, gr2) and torch.equal(gr1, dy) and torch.equal(gg1, gg2)


@pytest.mark.parametrize("dt,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)], ids=["f32", "bf16"])
def test_spectral_norm_conv_scale_in_the_epilogue(hific, dev, dt, tol):
    """flags bit 4 (round 4): 1/sigma multiplies the accumulator in the conv epilogue and the packed weights are those of
    weight_orig alone (cached across forwards) - against the scaled-pack form and against torch on W / sigma."""
    from hific_amd import ops, lib
    hific.set_compute_dtype(dt)
    ops.pack_cache.clear()
    x = _rnd((4, 64, 32, 32), 1).to(dev).to(dt)
    w = (_rnd((128, 64, 4, 4), 2) * 0.05).to(dev)
    bias = _rnd((128,), 3).to(dev)
    u = F.normalize(_rnd((128,), 4), dim=0).to(dev)
    v = F.normalize(_rnd((1024,), 5), dim=0).to(dev)
    geom = (2, 1, 1, 1, 1, lib.PAD_REFLECT)
    res = {}
    was = ops._SN_EPI_SCALE
    try:
        for mode in (16, 0):
            ops._SN_EPI_SCALE = mode
            xd = x.clone().requires_grad_(True)
            wd = w.clone().requires_grad_(True)
            bd = bias.clone().requires_grad_(True)
            uu, vv = u.clone(), v.clone()
            sig = ops.spectral_norm_power_iteration(wd.detach(), uu, vv, do_iter=True)
            y = ops.SNConv2dFn.apply(xd, wd, bd, uu, vv, sig, geom, "leaky_relu", False)
            y.float().square().sum().backward()
            torch.cuda.synchronize()
            res[mode] = (y.detach().float().cpu(), xd.grad.float().cpu(), wd.grad.cpu(), bd.grad.cpu(), float(sig[0]))
    finally:
        ops._SN_EPI_SCALE = was
        ops.pack_cache.clear()
        hific.set_compute_dtype(torch.float32)
    for a, b2 in zip(res[16][:4], res[0][:4]):
        assert _relerr(a, b2) < tol
    xp = F.pad(x.float().cpu(), (1, 1, 1, 1), mode="reflect")
    yr = F.leaky_relu(F.conv2d(xp, w.cpu() / res[16][4], bias.cpu(), stride=2), 0.2)
    assert _relerr(res[16][0], yr) < tol

"""GPU parity of the drop-in modules against the oracle restatement of the reference (same state_dict, same
inputs, same injected noise): forward, input gradients and weight gradients, float32 parity mode (<=1e-3 relative,
bit-exact quantised latent indices) and bf16 mode (looser, reported)."""
import pytest
import torch

from oracle import hific_oracle as O
from gradcheck import check_grads

pytestmark = pytest.mark.gpu

N_RES = 2


def _relerr(a, b):
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-20)


def _gtol(tol):
    """Gradient tolerance: float32 mode holds gradients to the forward bar (1e-3 of each tensor's scale, with the float64
    oracle as arbiter where the float32 oracle itself is noisier than that: tests/gradcheck.py); bf16 mode to 10x its
    forward tolerance (two bf16-rounded operands per product, sums of 10^4..10^6 terms)."""
    return 1e-3 if tol <= 1e-3 else tol * 10


def _check_grads(got, ref32, tol, what, exact=None):
    """`exact`: callable -> {name: float64 oracle gradient}; only consulted in float32 mode for tensors beyond `tol`."""
    check_grads(got, ref32, exact if tol <= 1e-3 else None, tol, what)


@pytest.fixture(scope="module")
def sd():
    return O.make_state_dict(seed=0, gan=True, n_res=N_RES)


def _load(module, sd, prefix):
    sub = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    module.load_state_dict(sub, strict=True)
    return module


@pytest.mark.parametrize("dt,tol", [(torch.float32, 1e-3), (torch.bfloat16, 6e-2)], ids=["f32", "bf16"])
def test_encoder(hific, dev, sd, dt, tol):
    from hific_amd.network.encoder import Encoder
    hific.set_compute_dtype(dt)
    enc = _load(Encoder((3, 128, 128), 2, C=220), sd, "Encoder.").to(dev)
    x = O.make_image(1, 2, 128, 128)
    # (conv_block2.1.bias: produced by the ChannelNorm backward kernel, normalisation.channel.fuse_bias_grad)
    watch = ("conv_block1.1.weight", "conv_block3.1.weight", "conv_block5.2.gamma", "conv_block_out.1.bias",
             "conv_block2.1.bias", "conv_block5.1.bias")

    def oracle(odt):
        sdr = {k: v.to(odt).clone().requires_grad_(True) for k, v in sd.items() if k.startswith("Encoder.")}
        yr = O.encoder_forward(sdr, x.to(odt))
        g = O.make_noise(2, tuple(yr.shape)) * 2
        yr.backward(g.to(odt))
        return yr.detach(), g, {k: sdr["Encoder." + k].grad for k in watch}

    yr, g, ref = oracle(torch.float32)
    y = enc(x.to(dev))
    assert y.dtype == torch.float32
    y.backward(g.to(dev))
    torch.cuda.synchronize()
    assert _relerr(y.detach().cpu(), yr) < tol
    params = dict(enc.named_parameters())
    _check_grads({k: params[k].grad.cpu() for k in watch}, ref, _gtol(tol), f"Encoder {dt}", lambda: oracle(torch.float64)[2])


@pytest.mark.parametrize("dt,tol", [(torch.float32, 1e-3), (torch.bfloat16, 6e-2)], ids=["f32", "bf16"])
def test_generator(hific, dev, sd, dt, tol):
    from hific_amd.network.generator import Generator
    hific.set_compute_dtype(dt)
    gen = _load(Generator((3, 128, 128), 2, C=220, n_residual_blocks=N_RES), sd, "Generator.").to(dev)
    y = O.make_noise(3, (2, 220, 8, 8)) * 4
    watch = ("resblock_0.conv1.weight", "upconv_block2.0.weight", "conv_block_out.1.weight", "resblock_1.norm2.beta",
             "resblock_0.conv1.bias", "resblock_1.conv2.bias", "upconv_block2.0.bias", "conv_block_init.2.bias")

    def oracle(odt):
        sdr = {k: v.to(odt).clone().requires_grad_(True) for k, v in sd.items() if k.startswith("Generator.")}
        yr_in = y.to(odt).clone().requires_grad_(True)
        xr = O.generator_forward(sdr, yr_in, N_RES)
        g = O.make_noise(4, tuple(xr.shape))
        xr.backward(g.to(odt))
        grads = {k: sdr["Generator." + k].grad for k in watch}
        grads["input"] = yr_in.grad
        return xr.detach(), g, grads

    xr, g, ref = oracle(torch.float32)
    yd = y.to(dev).requires_grad_(True)
    x = gen(yd)
    x.backward(g.to(dev).to(x.dtype))
    torch.cuda.synchronize()
    assert _relerr(x.detach().float().cpu(), xr) < tol
    params = dict(gen.named_parameters())
    got = {k: params[k].grad.cpu() for k in watch}
    got["input"] = yd.grad.cpu()
    _check_grads(got, ref, _gtol(tol), f"Generator {dt}", lambda: oracle(torch.float64)[2])


@pytest.mark.parametrize("training", [True, False])
def test_hyperprior_fp32(hific, dev, sd, training):
    from hific_amd.hyperprior import Hyperprior
    hific.set_compute_dtype(torch.float32)
    hp = _load(Hyperprior(bottleneck_capacity=220), sd, "Hyperprior.").to(dev).train(training)
    y = O.make_noise(5, (2, 220, 8, 8)) * 6
    nh, nl = O.make_noise(6, (2, 320, 2, 2)), O.make_noise(7, (2, 220, 8, 8))
    gdec = O.make_noise(8, (2, 220, 8, 8))
    watch = ("analysis_net.conv1.weight", "synthesis_mu.conv2.weight", "synthesis_std.conv3.bias",
             "hyperlatent_likelihood.H_1", "hyperlatent_likelihood.a_0", "hyperlatent_likelihood.b_3")

    def oracle(odt, symbols=None):
        sdr = {k: v.to(odt).clone().requires_grad_(True) for k, v in sd.items() if k.startswith("Hyperprior.")}
        yr = y.to(odt).clone().requires_grad_(True)
        hr = O.hyperprior_forward(sdr, yr, (128, 128), training, nh.to(odt), nl.to(odt), symbols_override=symbols)
        (hr.total_nbpp * 3.0 + hr.total_qbpp * 0.5 + (hr.decoded * gdec.to(odt)).sum()).backward()
        grads = {k: sdr["Hyperprior." + k].grad for k in watch}
        grads["latents"] = yr.grad
        return hr, grads

    hr, ref = oracle(torch.float32)
    noises = [nh.to(dev), nl.to(dev)]
    hp._draw_noise = lambda t: noises.pop(0)
    yd = y.to(dev).requires_grad_(True)
    h = hp(yd, (128, 128))
    (h.total_nbpp * 3.0 + h.total_qbpp * 0.5 + (h.decoded * gdec.to(dev)).sum()).backward()
    torch.cuda.synchronize()
    for f in ("latent_nbpp", "hyperlatent_nbpp", "total_nbpp", "latent_qbpp", "hyperlatent_qbpp", "total_qbpp"):
        a, b = float(getattr(h, f)), float(getattr(hr, f))
        assert abs(a - b) < 1e-3 * abs(b), (f, a, b)
    # quantised indices: exact, except where the oracle's own y - mu sits within f32 noise of a rounding tie
    # (decoded = index + mu, so round(decoded - mu_oracle) recovers the HIP path's index)
    idx_o = O.quantized_indices(y, hr.latent_means.detach())
    idx_h = torch.round(h.decoded.detach().cpu() - hr.latent_means.detach()).to(torch.int64)
    flips = idx_h != idx_o
    if flips.any():
        frac = y - hr.latent_means.detach() + 0.5
        frac = frac - torch.floor(frac)
        tie = torch.minimum(frac, 1 - frac)
        assert int(flips.sum()) <= 2 and float(tie[flips].max()) < 1e-4, (int(flips.sum()), float(tie[flips].max()))
        assert int(((idx_h - idx_o).abs() > 1).sum()) == 0
    params = dict(hp.named_parameters())
    got = {k: params[k].grad.cpu() for k in watch}
    got["latents"] = yd.grad.cpu()
    sym = idx_h.to(torch.float32) if flips.any() else None
    if sym is not None:
        ref = oracle(torch.float32, sym)[1]                  # "given equal indices"
    _check_grads(got, ref, 1e-3, f"Hyperprior f32 training={training}", lambda: oracle(torch.float64, sym)[1])


@pytest.mark.parametrize("dt,tol", [(torch.float32, 1e-3), (torch.bfloat16, 5e-2)], ids=["f32", "bf16"])
def test_discriminator(hific, dev, sd, dt, tol):
    from hific_amd.network.discriminator import Discriminator
    hific.set_compute_dtype(dt)
    D = _load(Discriminator((3, 128, 128), (220, 8, 8), C=220), sd, "Discriminator.").to(dev).train()
    x = O.make_image(9, 4, 128, 128)
    y = O.make_noise(10, (4, 220, 8, 8)) * 4
    watch = ("conv1.weight_orig", "conv4.weight_orig", "conv3.bias", "context_conv.weight", "conv_out.weight")

    def oracle(odt):
        sdr = {k: (v.to(odt).clone().requires_grad_(True) if "weight_u" not in k and "weight_v" not in k else v.to(odt))
               for k, v in sd.items() if k.startswith("Discriminator.")}
        xr = x.to(odt).clone().requires_grad_(True)
        out_r, logit_r, new_uv = O.discriminator_forward(sdr, xr, y.to(odt), training=True)
        g = O.make_noise(11, tuple(logit_r.shape))
        logit_r.backward(g.to(odt))
        grads = {k: sdr["Discriminator." + k].grad for k in watch}
        grads["input"] = xr.grad
        return out_r.detach(), logit_r.detach(), new_uv, g, grads

    out_r, logit_r, new_uv, g, ref = oracle(torch.float32)
    xd = x.to(dev).requires_grad_(True)
    out, logits = D(xd, y.to(dev))
    logits.backward(g.to(dev))
    torch.cuda.synchronize()
    assert _relerr(logits.detach().cpu(), logit_r) < tol
    assert _relerr(out.cpu(), out_r) < tol
    assert _relerr(D.conv2.weight_u.cpu(), new_uv["Discriminator.conv2.weight_u"]) < 1e-4
    params = dict(D.named_parameters())
    got = {k: params[k].grad.cpu() for k in watch}
    got["input"] = xd.grad.cpu()
    _check_grads(got, ref, _gtol(tol), f"Discriminator {dt}", lambda: oracle(torch.float64)[4])


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_discriminator_pair_input_equals_cat_and_repeat_interleave(hific, dev, sd, dt):
    """Discriminator.forward_pairs(real, gen, latents) against forward(cat([real, gen]), repeat_interleave(latents, 2)) - the
    call of src/model.py:176-179: same outputs bit for bit (the context conv of a repeated latent is the conv of the latent),
    same gradients for the generated images; the context-conv gradients agree to rounding (the pair path adds the two
    images' context gradients before the weight gradient instead of inside it)."""
    from hific_amd.network.discriminator import Discriminator
    hific.set_compute_dtype(dt)
    D = _load(Discriminator((3, 128, 128), (220, 8, 8), C=220), sd, "Discriminator.").to(dev).eval()    # eval: no power iteration
    B = 3
    real = O.make_image(9, B, 128, 128).to(dev).to(dt)
    gen0 = O.make_image(12, B, 128, 128).to(dev).to(dt)
    lat = (O.make_noise(10, (B, 220, 8, 8)) * 4).to(dev)
    g = None
    res = {}
    from hific_amd import ops
    for mode in ("cat", "pair", "stage"):
        D.zero_grad()
        gen = gen0.clone().requires_grad_(True)
        if mode == "cat":
            out, logits = D(torch.cat([real, gen], dim=0), torch.repeat_interleave(lat, 2, dim=0))
        else:
            # "stage": input gather + first convolution as one node whose backward never forms the 15-channel data gradient
            ops.set_d1_stage(mode == "stage")
            try:
                out, logits = D.forward_pairs(real, gen, lat)
            finally:
                ops.set_d1_stage(True)
        if g is None:
            g = O.make_noise(11, tuple(logits.shape)).to(dev)
        logits.backward(g)
        torch.cuda.synchronize()
        res[mode] = (out.clone(), logits.detach().clone(), gen.grad.clone(),
                     {k: p.grad.clone() for k, p in D.named_parameters() if p.grad is not None})
    tol = 1e-5 if dt == torch.float32 else 2e-2
    for mode in ("pair", "stage"):
        assert torch.equal(res["cat"][0], res[mode][0]) and torch.equal(res["cat"][1], res[mode][1]), mode
        if mode == "pair":
            assert torch.equal(res["cat"][2], res[mode][2])
        else:       # the restricted data gradient runs the same contraction on a 3-row problem: equal up to its tiling
            assert _relerr(res[mode][2].float().cpu(), res["cat"][2].float().cpu()) < tol
        for k, gc in res["cat"][3].items():
            gp = res[mode][3][k]
            if not k.startswith("context_conv."):
                assert torch.equal(gc, gp), (mode, k)
            else:
                assert _relerr(gp.float().cpu(), gc.float().cpu()) < tol, (mode, k)
    hific.set_compute_dtype(torch.float32)


def test_discriminator_input_stage_is_chosen_only_where_its_backward_has_a_kernel(hific, dev, sd, monkeypatch):
    """ops.D1StageFn's backward (hific_d1_ctx_grad) needs >= 2 cells per side and 4 KiB of workspace per cell;
    forward_pairs must check that BEFORE choosing the fused node (it used to fail in the middle of backward) and otherwise take
    UpsamplePairConcatFn + SNConv2d, with the same results.  (A 1-row latent never reaches the Discriminator: its context
    convolution reflect-pads by 1, which torch rejects on a 1-row plane too.)"""
    from hific_amd.network.discriminator import Discriminator
    from hific_amd import lib
    hific.set_compute_dtype(torch.float32)
    D = _load(Discriminator((3, 32, 64), (220, 2, 4), C=220), sd, "Discriminator.").to(dev).eval()
    B = 2
    real = O.make_image(9, B, 32, 64).to(dev)
    gen0 = O.make_image(12, B, 32, 64).to(dev)
    lat = (O.make_noise(10, (B, 220, 2, 4)) * 4).to(dev)
    assert not D._d1_stage_eligible(torch.empty(B, 3, 16, 64, device=dev), torch.empty(B, 12, 1, 4, device=dev))
    assert D._d1_stage_eligible(real, torch.empty(B, 12, 2, 4, device=dev))
    res = {}
    for mode in ("fused", "small_workspace"):
        if mode == "small_workspace":           # not enough workspace for the window sums -> the two-node path
            small = torch.empty(B * 2 * 4 * 4096 - 1, dtype=torch.uint8, device=dev)
            monkeypatch.setattr(lib, "workspace", lambda device, min_bytes=0: small)
            assert not D._d1_stage_eligible(real, torch.empty(B, 12, 2, 4, device=dev))
            monkeypatch.undo()
            monkeypatch.setattr(D, "_d1_stage_eligible", lambda *a: False)
        D.zero_grad()
        gen = gen0.clone().requires_grad_(True)
        out, logits = D.forward_pairs(real, gen, lat)
        logits.sum().backward()
        torch.cuda.synchronize()
        res[mode] = (logits.detach().clone(), gen.grad.clone())
    assert torch.equal(res["fused"][0], res["small_workspace"][0])
    assert _relerr(res["fused"][1].cpu(), res["small_workspace"][1].cpu()) < 1e-5


@pytest.mark.parametrize("shape", [(2, 64, 64), (3, 32, 512), (1, 256, 256)], ids=["64x64", "32x512_segments", "256x256"])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_context_gradient_through_first_discriminator_conv(hific, dev, shape, dt):
    """hific_d1_ctx_grad against autograd through cat(image, upsample16(context)) -> reflect pad 1 -> 4x4 stride-2 convolution
    (src/network/discriminator.py:36,53,75-78) with the reference's latent pairing (image i reads context i >> 1,
    src/model.py:176-179): first / last block rows and columns (mirror lines), several column segments, both dtypes."""
    import torch.nn.functional as F
    from hific_amd import lib
    B, H, W = shape
    Ci, Cc, K, f = 3, 12, 64, 16
    g = torch.Generator().manual_seed(5)
    w = (torch.rand((K, Ci + Cc, 4, 4), generator=g) - 0.5) * 0.2
    dz = (torch.rand((2 * B, K, H // 2, W // 2), generator=g) - 0.5)
    if dt == torch.bfloat16:
        dz = dz.bfloat16().float()
    inv_sigma = torch.tensor([0.7])
    ctx = torch.zeros((B, Cc, H // f, W // f), requires_grad=True)
    img = torch.zeros((2 * B, Ci, H, W))
    up = F.interpolate(torch.repeat_interleave(ctx, 2, dim=0), scale_factor=f, mode="nearest")
    y = F.conv2d(F.pad(torch.cat([img, up], 1), (1, 1, 1, 1), mode="reflect"), w * inv_sigma, stride=2)
    (y * dz).sum().backward()
    out = torch.empty((B, Cc, H // f, W // f), dtype=dt, device=dev)
    dzd = dz.to(dev).to(dt).contiguous()
    wd, sd_ = w.to(dev).contiguous(), inv_sigma.to(dev)
    ws = lib.workspace(dev)
    lib.call("hific_d1_ctx_grad", dzd.data_ptr(), wd.data_ptr(), sd_.data_ptr(), out.data_ptr(), B, K, Ci, Cc, H, W, f,
             lib.dtype_code(dzd), ws.data_ptr(), ws.numel(), lib.stream())
    torch.cuda.synchronize()
    err = _relerr(out.float().cpu(), ctx.grad)
    print(f"context gradient {shape} {dt}: max-rel {err:.2e}")
    assert err < (2e-5 if dt == torch.float32 else 6e-3)      # bf16: the output rounding only (sums in float32)


@pytest.mark.parametrize("gan", [False, True], ids=["compression", "compression_gan"])
def test_model_losses_and_grads_fp32(hific, dev, sd, gan):
    """End-to-end training forward/backward (config 1 semantics at reduced size): losses within 1e-3 of the oracle,
    reconstruction within 1e-3, selected gradients within 1e-2."""
    import hific_amd
    from hific_amd.default_config import make_args, mse_lpips_args, hific_args, ModelTypes
    hific.set_compute_dtype(torch.float32)
    args = make_args(hific_args if gan else mse_lpips_args, n_residual_blocks=N_RES)
    model = hific_amd.Model(args, model_type=ModelTypes.COMPRESSION_GAN if gan else ModelTypes.COMPRESSION,
                            allow_random_lpips_backbone=True)
    model.load_state_dict({k: v for k, v in sd.items() if gan or not k.startswith("Discriminator.")}, strict=True)
    bb = O.make_alex_backbone()
    model.perceptual_loss.load_backbone_state_dict(bb)
    model = model.to(dev).train()
    lins = [getattr(model.perceptual_loss, f"lin{i}").cpu() for i in range(5)]
    nh, nl = O.make_noise(6, (2, 320, 2, 2)), O.make_noise(7, (2, 220, 8, 8))
    # pick an input whose latents are not within f32 noise of a rounding tie (so index equality is well-posed)
    for img_seed in range(1, 12):
        x = O.make_image(img_seed, 2, 128, 128)
        with torch.no_grad():
            y0 = O.encoder_forward(sd, x)
            h0 = O.hyperprior_forward(sd, y0, (128, 128), True, nh, nl)
            fr = y0 - h0.latent_means + 0.5
            fr = fr - torch.floor(fr)
        if float(torch.minimum(fr, 1 - fr).min()) > 2e-5:
            break
    noises = [nh.to(dev), nl.to(dev)]
    model.Hyperprior._draw_noise = lambda t: noises.pop(0)
    losses, inter = model(x.to(dev), train_generator=True, return_intermediates=True, writeout=False)
    losses["compression"].backward()
    torch.cuda.synchronize()
    watch = ["Encoder.conv_block2.1.weight", "Generator.resblock_0.conv2.weight", "Generator.upconv_block4.0.weight",
             "Hyperprior.synthesis_std.conv1.weight", "Hyperprior.hyperlatent_likelihood.H_2"]
    if gan:
        watch.append("Discriminator.conv2.weight_orig")
    def oracle(odt):
        sdr = {k: ((v.to(odt) if v.dtype.is_floating_point else v).clone().requires_grad_(k in watch)) for k, v in sd.items()}
        out = O.model_forward(sdr, {k: v.to(odt) for k, v in bb.items()}, [l.to(odt) for l in lins], x.to(odt),
                              step_counter=1, training=True, gan=gan, train_generator=True, noise_hyper=nh.to(odt),
                              noise_latent=nl.to(odt), n_residual_blocks=N_RES)
        out["compression"].backward()
        return out, {k: sdr[k].grad for k in watch}

    out, ref = oracle(torch.float32)
    a, b = float(losses["compression"]), float(out["compression"])
    assert abs(a - b) < 1e-3 * abs(b), (a, b)
    if gan:
        assert abs(float(losses["disc"]) - float(out["disc"])) < 1e-3 * abs(float(out["disc"]))
    # quantised latents: integer-exact, except where the oracle's own value sits on a rounding tie (|frac-.5|<1e-4:
    # f32 summation-order noise decides those on either side); the reconstruction is then checked against the
    # oracle Generator run on the HIP path's latents
    hi = out["hyperinfo"]
    dec_h, dec_o = inter.latents_quantized.detach().cpu(), hi.decoded.detach()
    flips = (dec_h - dec_o).abs() > 1e-3
    rec_ref = out["reconstruction"].detach()
    if flips.any():
        frac = (out["y"].detach() - hi.latent_means.detach() + 0.5)
        frac = frac - torch.floor(frac)
        tie = torch.minimum(frac, 1 - frac)
        print(f"rounding flips: {int(flips.sum())} of {flips.numel()}, max tie distance {float(tie[flips].max()):.2e}")
        assert int(flips.sum()) <= 4 and float(tie[flips].max()) < 1e-4
        rec_ref = O.generator_forward(sd, dec_h, N_RES)
    assert _relerr(inter.reconstruction.detach().float().cpu(), rec_ref) < 1e-3
    params = dict(model.named_parameters())
    _check_grads({k: params[k].grad.cpu() for k in watch}, ref, 1e-3, f"Model f32 gan={gan}",
                 lambda: oracle(torch.float64)[1])


def test_zz_hyperprior_compress_roundtrip(hific, dev, tmp_path):
    """compress_forward -> .hfc -> decompress_forward on the device modules: the decoder reproduces the latents
    quantised around the means that the (same) synthesis kernels predict from the decoded hyperlatents."""
    import hific_amd
    from hific_amd.compression import container
    hific.set_compute_dtype(torch.float32)
    torch.manual_seed(0)
    # scalar coder: lossless for any symbol (the vectorised default is lossy on multi-nibble overflows, like the reference)
    hp = hific_amd.hyperprior.Hyperprior(bottleneck_capacity=24, hyperlatent_filters=32,
                                         vectorize_encoding=False).to(dev).eval()
    hp.build_tables()
    y = (O.make_noise(3, (1, 24, 16, 16)) * 3).to(dev)
    out = hp.compress_forward(y, (256, 256))
    path = str(tmp_path / "gpu.hfc")
    container.save_compressed_format(out, path)
    y_hat = hp.decompress_forward(container.load_compressed_format(path), dev)
    with torch.no_grad():
        z_hat = torch.floor(hp.analysis_net(y) + 0.5)
        mu = hp.synthesis_mu(z_hat)
    torch.cuda.synchronize()
    assert y_hat.shape == y.shape
    assert torch.equal(y_hat.cpu(), (torch.floor(y + 0.5 - mu) + mu).cpu())


def test_zz_model_compress_decompress(hific, dev, tmp_path):
    """Model.compress -> .hfc -> Model.decompress at a size that needs both paddings (image 72x88 -> 80x96, latents
    5x6 -> 8x8): reconstruction has the image's size, lies in [0,1] and equals the Generator run on the decoded latents."""
    import hific_amd
    from hific_amd.compression import container
    from hific_amd.default_config import make_args, mse_lpips_args, ModelTypes, ModelModes
    hific.set_compute_dtype(torch.float32)
    torch.manual_seed(0)
    args = make_args(mse_lpips_args, n_residual_blocks=1)
    model = hific_amd.Model(args, model_type=ModelTypes.COMPRESSION, model_mode=ModelModes.EVALUATION,
                            allow_random_lpips_backbone=True).to(dev).eval()
    model.Hyperprior.vectorize_encoding = False
    model.Hyperprior.build_tables()
    x = O.make_image(9, 1, 72, 88).to(dev)
    out = model.compress(x)
    assert tuple(out.spatial_shape) == (72, 88)
    # the reference's reporting fields (Shannon estimates from the likelihood kernels) next to the attained size
    attained = 32.0 * (len(out.hyperlatents_encoded) + len(out.latents_encoded))
    assert out.total_bits > 0 and abs(out.total_bpp - out.total_bits / (72 * 88)) < 1e-6
    assert 0.3 < out.total_bits / attained < 3.0, (out.total_bits, attained)
    # EVALUATION-mode forward (model.py:357-366): clamped reconstruction and the quantised rate
    rec_f, q_bpp = model(O.make_image(10, 1, 128, 160).to(dev))
    assert tuple(rec_f.shape) == (1, 3, 128, 160) and float(rec_f.min()) >= 0.0 and float(rec_f.max()) <= 1.0
    assert float(q_bpp) > 0
    path = str(tmp_path / "img.hfc")
    container.save_compressed_format(out, path)
    rec = model.decompress(container.load_compressed_format(path))
    torch.cuda.synchronize()
    assert tuple(rec.shape) == (1, 3, 72, 88) and float(rec.min()) >= 0.0 and float(rec.max()) <= 1.0
    with torch.no_grad():
        lat = model.Hyperprior.decompress_forward(out, device=dev)
        ref = torch.clamp(model.Generator(lat)[:, :, :72, :88].float(), 0.0, 1.0)
    assert torch.equal(rec.cpu(), ref.cpu())


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_instance_norm_variant_trains(hific, dev, dt):
    """use_channel_norm=False: HIP convs + the PyTorch instance-norm fallback, forward and backward on the device,
    against the same network assembled from torch ops on the CPU."""
    import torch.nn.functional as F
    from hific_amd.network.encoder import Encoder
    hific.set_compute_dtype(dt)
    torch.manual_seed(0)
    enc = Encoder((3, 64, 64), 2, C=220, channel_norm=False)
    sd = {k: v.clone() for k, v in enc.state_dict().items()}
    x = O.make_image(1, 2, 64, 64)
    enc = enc.to(dev)
    y = enc(x.to(dev))
    y.float().sum().backward()
    torch.cuda.synchronize()
    # CPU restatement of encoder.py with InstanceNorm (pads: 3 / asymmetric (0,1,1,0) / 1, all reflect)
    h = x
    for i, (pad, st) in enumerate([((3, 3, 3, 3), 1)] + [((0, 1, 1, 0), 2)] * 4, start=1):
        h = F.conv2d(F.pad(h, pad, mode="reflect"), sd[f"conv_block{i}.1.weight"], sd[f"conv_block{i}.1.bias"], stride=st)
        h = F.relu(F.instance_norm(h, weight=sd[f"conv_block{i}.2.weight"], bias=sd[f"conv_block{i}.2.bias"]))
    h = F.conv2d(F.pad(h, (1, 1, 1, 1), mode="reflect"), sd["conv_block_out.1.weight"], sd["conv_block_out.1.bias"])
    assert _relerr(y.detach().float().cpu(), h) < (1e-3 if dt == torch.float32 else 6e-2)
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in enc.parameters())


@pytest.mark.gpu
def test_branch_streams_equal_single_stream(hific, dev, sd):
    """model.py runs the distortion / LPIPS / rate branch on a second stream next to the Discriminator branch (and, with
    arenas, weight gradients on a third): every loss and every parameter gradient of a G-turn must be bit-identical to
    single-stream execution."""
    import hific_amd
    from hific_amd import ops, optim
    from hific_amd.default_config import make_args, hific_args, ModelTypes
    hific.set_compute_dtype(torch.bfloat16)

    def run(streams_on):
        ops.set_branch_streams(streams_on); ops.set_side_stream(streams_on)
        torch.manual_seed(0)
        model = hific_amd.Model(make_args(hific_args, n_residual_blocks=N_RES), model_type=ModelTypes.COMPRESSION_GAN,
                                allow_random_lpips_backbone=True)
        model.load_state_dict(sd, strict=True)
        model = model.to(dev).train()
        amort = optim.FusedAdam([p for m in model.amortization_models for p in m.parameters()], lr=1e-4)
        disc = optim.FusedAdam(list(model.Discriminator.parameters()), lr=1e-4)
        x = O.make_image(3, 4, 128, 128).to(dev)
        torch.manual_seed(1)
        losses = model(x, train_generator=True, writeout=False)
        losses["compression"].backward()
        out = (float(losses["compression"]), float(losses["disc"]), amort.arena.flat_grad.clone(), disc.arena.flat_grad.clone())
        torch.cuda.synchronize()
        return out

    was = (ops.branch_streams_on(), ops._SIDE_ON)
    try:
        a = run(False)
        b = run(True)
    finally:
        ops.set_branch_streams(was[0]); ops.set_side_stream(was[1])
    assert a[0] == b[0] and a[1] == b[1]
    assert torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])


def test_generator_sample_noise_variant(hific, dev):
    """Generator(sample_noise=True) (generator.py:105-107, 149-152: 32 noise channels concatenated to the head, 992-channel
    residual blocks): forward and input gradient vs the oracle with the same draw."""
    from hific_amd.network.generator import Generator
    hific.set_compute_dtype(torch.float32)
    torch.manual_seed(11)
    gen = Generator((8, 16, 16), 2, C=8, n_residual_blocks=1, sample_noise=True, noise_dim=32).to(dev)
    assert gen.resblock_0.conv1.weight.shape[:2] == (992, 992)
    sdg = {"Generator." + k: v.detach().cpu() for k, v in gen.state_dict().items()}
    y = O.make_noise(21, (2, 8, 16, 16))
    z = O.make_noise(22, (2, 32, 16, 16))
    gen._draw_noise = lambda shape: z.clone()
    yr = y.clone().requires_grad_(True)
    out_or = O.generator_forward(sdg, yr, 1, noise=z)
    out_or.square().mean().backward()
    yd = y.to(dev).requires_grad_(True)
    out = gen(yd)
    out.float().square().mean().backward()
    torch.cuda.synchronize()
    assert out.shape == (2, 3, 256, 256)
    assert _relerr(out.detach().float().cpu(), out_or.detach()) < 1e-3
    assert _relerr(yd.grad.float().cpu(), yr.grad) < 1e-2


def test_normalize_input_image_variant(hific, dev, sd):
    """args.normalize_input_image=True (model.py:155-156, 206-209, 361-363): tanh on the reconstruction, [-1,1] -> [0,1]
    before the losses.  Losses and a gradient vs the oracle (pinned to the reference's Model in
    tests/test_oracle_vs_reference.py::test_normalize_input_image_variant)."""
    import hific_amd
    from hific_amd.default_config import make_args, mse_lpips_args, ModelTypes, ModelModes
    hific.set_compute_dtype(torch.float32)
    args = make_args(mse_lpips_args, n_residual_blocks=N_RES, normalize_input_image=True)
    model = hific_amd.Model(args, model_type=ModelTypes.COMPRESSION, allow_random_lpips_backbone=True)
    model.load_state_dict({k: v for k, v in sd.items() if not k.startswith("Discriminator.")}, strict=True)
    bb = O.make_alex_backbone()
    model.perceptual_loss.load_backbone_state_dict(bb)
    model = model.to(dev).train()
    lins = [getattr(model.perceptual_loss, f"lin{i}").cpu() for i in range(5)]
    nh, nl = O.make_noise(6, (2, 320, 2, 2)), O.make_noise(7, (2, 220, 8, 8))
    x = O.make_image(2, 2, 128, 128) * 2 - 1
    noises = [nh.to(dev), nl.to(dev)]
    model.Hyperprior._draw_noise = lambda t: noises.pop(0)
    losses = model(x.to(dev), train_generator=True, writeout=False)
    losses["compression"].backward()
    torch.cuda.synchronize()
    key = "Generator.conv_block_out.1.weight"
    sdr = {k: (v.clone().requires_grad_(True) if k == key else v.clone()) for k, v in sd.items()}
    out = O.model_forward(sdr, bb, lins, x, step_counter=1, training=True, gan=False, noise_hyper=nh, noise_latent=nl,
                          args=dict(normalize_input_image=True), n_residual_blocks=N_RES)
    out["compression"].backward()
    a, b = float(losses["compression"]), float(out["compression"])
    assert abs(a - b) < 1e-3 * abs(b), (a, b)
    g = dict(model.named_parameters())[key].grad.float().cpu()
    assert _relerr(g, sdr[key].grad) < 1e-2
    # EVALUATION forward maps the tanh output to [0, 1]
    del model.Hyperprior._draw_noise                 # back to the RNG draw
    model.eval(); model.model_mode = ModelModes.EVALUATION
    with torch.no_grad():
        rec, _ = model(x.to(dev))
    assert float(rec.min()) >= 0.0 and float(rec.max()) <= 1.0

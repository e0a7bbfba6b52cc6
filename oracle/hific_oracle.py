"""ORACLE — TEST INFRASTRUCTURE ONLY.  CPU restatement of the reference's HiFIC training/validation hot path.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module; the product
package (hific_amd) never does.  It is a plain-PyTorch (CPU, float32 or float64) functional restatement of what the
reference computes, written from the behaviour, one function per reference call site, each
citing the file:line it follows.  Everything takes an explicit `state_dict` whose keys are the reference's
(`Encoder.conv_block1.1.weight`, ...), so the same tensors can be loaded into the reference, the oracle and the
HIP modules.

Pinning: tests/test_oracle_vs_reference.py checks every function here against the *imported reference modules*
(oracle/ref_loader.py, only where /root/reference exists) and tests/golden/*.pt hold outputs of the reference itself
on seeded inputs (made by tests/golden/make_golden.py) which tests/test_oracle_golden.py replays anywhere.
"""
import math
from collections import namedtuple

import numpy as np
import torch
import torch.nn.functional as F

HyperInfo = namedtuple("HyperInfo", "decoded latent_nbpp hyperlatent_nbpp total_nbpp latent_qbpp hyperlatent_qbpp "
                                    "total_qbpp latent_means latent_scales hyperlatents")

MIN_SCALE = 0.11
MIN_LIKELIHOOD = 1e-9


# ---- primitives ---------------------------------------------------------------------------------------
def channel_norm(x, gamma, beta, eps=1e-3):
    """src/normalisation/channel.py:48-59 — moments over dim=1, *unbiased* variance, eps inside rsqrt."""
    mu = x.mean(dim=1, keepdim=True)
    var = x.var(dim=1, keepdim=True)            # unbiased (C-1)
    return gamma * ((x - mu) * torch.rsqrt(var + eps)) + beta


def reflect_pad(x, left, right, top, bottom):
    """nn.ReflectionPad2d((l, r, t, b)) — src/network/encoder.py:46-48, generator.py:27,86-88."""
    return F.pad(x, (left, right, top, bottom), mode="reflect")


class _LowerBoundToward(torch.autograd.Function):
    """src/helpers/maths.py:87-100 — clamp; gradient passes where x>=bound or grad<0."""

    @staticmethod
    def forward(ctx, x, bound):
        ctx.save_for_backward(x.ge(bound))
        return torch.clamp(x, min=bound)

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        return g * torch.logical_or(mask, g.lt(0.)).to(g.dtype), None


def lower_bound_toward(x, bound):
    return _LowerBoundToward.apply(x, bound)


def std_cdf(v, likelihood_type="gaussian"):
    """src/helpers/maths.py:102-109."""
    if likelihood_type == "gaussian":
        return 0.5 * torch.erfc(v * (-1. / math.sqrt(2.)))
    return torch.sigmoid(v)


# ---- Encoder / Generator ------------------------------------------------------------------------------
def encoder_forward(sd, x, prefix="Encoder."):
    """src/network/encoder.py:56-111: 7x7 s1 (reflect 3) -> 4x [asym reflect (0,1,1,0), 3x3 s2] each followed by
    ChannelNorm + ReLU -> reflect 1, 3x3 s1 to C channels."""
    p = prefix
    h = F.conv2d(reflect_pad(x, 3, 3, 3, 3), sd[p + "conv_block1.1.weight"], sd[p + "conv_block1.1.bias"])
    h = F.relu(channel_norm(h, sd[p + "conv_block1.2.gamma"], sd[p + "conv_block1.2.beta"]))
    for i in range(2, 6):
        h = F.conv2d(reflect_pad(h, 0, 1, 1, 0), sd[p + f"conv_block{i}.1.weight"], sd[p + f"conv_block{i}.1.bias"],
                     stride=2)
        h = F.relu(channel_norm(h, sd[p + f"conv_block{i}.2.gamma"], sd[p + f"conv_block{i}.2.beta"]))
    return F.conv2d(reflect_pad(h, 1, 1, 1, 1), sd[p + "conv_block_out.1.weight"], sd[p + "conv_block_out.1.bias"])


def residual_block_forward(sd, x, prefix):
    """src/network/generator.py:31-44."""
    r = F.conv2d(reflect_pad(x, 1, 1, 1, 1), sd[prefix + "conv1.weight"], sd[prefix + "conv1.bias"])
    r = F.relu(channel_norm(r, sd[prefix + "norm1.gamma"], sd[prefix + "norm1.beta"]))
    r = F.conv2d(reflect_pad(r, 1, 1, 1, 1), sd[prefix + "conv2.weight"], sd[prefix + "conv2.bias"])
    r = channel_norm(r, sd[prefix + "norm2.gamma"], sd[prefix + "norm2.beta"])
    return r + x


def generator_forward(sd, y, n_residual_blocks=9, prefix="Generator.", noise=None):
    """src/network/generator.py:98-103,145-168.  `noise` (B, noise_dim, H, W): the sample_noise=True variant, whose
    draw is concatenated to the head (:149-152)."""
    p = prefix
    h = channel_norm(y, sd[p + "conv_block_init.0.gamma"], sd[p + "conv_block_init.0.beta"])
    h = F.conv2d(reflect_pad(h, 1, 1, 1, 1), sd[p + "conv_block_init.2.weight"], sd[p + "conv_block_init.2.bias"])
    head = channel_norm(h, sd[p + "conv_block_init.3.gamma"], sd[p + "conv_block_init.3.beta"])
    if noise is not None:
        head = torch.cat((head, noise.to(head)), dim=1)
    h = head
    for m in range(n_residual_blocks):
        h = residual_block_forward(sd, h, p + f"resblock_{m}.")
    h = h + head
    for i in range(1, 5):
        h = F.conv_transpose2d(h, sd[p + f"upconv_block{i}.0.weight"], sd[p + f"upconv_block{i}.0.bias"], stride=2,
                               padding=1, output_padding=1)
        h = F.relu(channel_norm(h, sd[p + f"upconv_block{i}.1.gamma"], sd[p + f"upconv_block{i}.1.beta"]))
    return F.conv2d(reflect_pad(h, 3, 3, 3, 3), sd[p + "conv_block_out.1.weight"], sd[p + "conv_block_out.1.bias"])


# ---- hyperprior ---------------------------------------------------------------------------------------
def hyper_analysis_forward(sd, x, prefix="Hyperprior.analysis_net."):
    """src/network/hyper.py:52-63: 3x3 s1 zero-pad 1, 5x5 s2 reflect 2, 5x5 s2 reflect 2."""
    p = prefix
    h = F.relu(F.conv2d(x, sd[p + "conv1.weight"], sd[p + "conv1.bias"], padding=1))
    h = F.relu(F.conv2d(reflect_pad(h, 2, 2, 2, 2), sd[p + "conv2.weight"], sd[p + "conv2.bias"], stride=2))
    return F.conv2d(reflect_pad(h, 2, 2, 2, 2), sd[p + "conv3.weight"], sd[p + "conv3.bias"], stride=2)


def hyper_synthesis_forward(sd, x, prefix):
    """src/network/hyper.py:83-97."""
    p = prefix
    h = F.relu(F.conv_transpose2d(x, sd[p + "conv1.weight"], sd[p + "conv1.bias"], stride=2, padding=2,
                                  output_padding=1))
    h = F.relu(F.conv_transpose2d(h, sd[p + "conv2.weight"], sd[p + "conv2.bias"], stride=2, padding=2,
                                  output_padding=1))
    return F.conv_transpose2d(h, sd[p + "conv3.weight"], sd[p + "conv3.bias"], stride=1, padding=1)


def factorized_cdf_logits(sd, x, prefix="Hyperprior.hyperlatent_likelihood."):
    """src/compression/hyperprior_model.py:305-326; x is (C,1,L)."""
    logits = x
    for k in range(4):
        H, a, b = sd[prefix + f"H_{k}"], sd[prefix + f"a_{k}"], sd[prefix + f"b_{k}"]
        logits = torch.bmm(F.softplus(H), logits) + b
        logits = logits + torch.tanh(a) * torch.tanh(logits)
    return logits


def factorized_likelihood(sd, z, prefix="Hyperprior.hyperlatent_likelihood."):
    """src/compression/hyperprior_model.py:349-384."""
    N, C, H, W = z.shape
    v = z.permute(1, 0, 2, 3).reshape(C, 1, -1)
    up = factorized_cdf_logits(sd, v + 0.5, prefix)
    lo = factorized_cdf_logits(sd, v - 0.5, prefix)
    sign = -torch.sign(up + lo).detach()
    lik = torch.abs(torch.sigmoid(sign * up) - torch.sigmoid(sign * lo))
    lik = lower_bound_toward(lik, MIN_LIKELIHOOD)
    return lik.reshape(C, N, H, W).permute(1, 0, 2, 3)


def estimate_entropy(lik, spatial_shape):
    """src/hyperprior.py:80-93."""
    B = lik.shape[0]
    n_bits = torch.sum(torch.log(lik + 1e-9)) / (B * (-math.log(2.)))
    return n_bits, n_bits / float(np.prod(spatial_shape))


def latent_likelihood(x, mean, scale, likelihood_type="gaussian"):
    """src/hyperprior.py:124-139."""
    a = torch.abs(x - mean)
    p = std_cdf((0.5 - a) / scale, likelihood_type) - std_cdf(-(0.5 + a) / scale, likelihood_type)
    return lower_bound_toward(p, MIN_LIKELIHOOD)


def hyperprior_forward(sd, latents, spatial_shape, training=True, noise_hyper=None, noise_latent=None,
                       likelihood_type="gaussian", prefix="Hyperprior.", symbols_override=None):
    """src/hyperprior.py:277-330.  The two uniform(-.5,.5) noise tensors are explicit arguments (the reference
    draws them from the global RNG, hyperlatent noise first: hyperprior.py:283,305).
    `symbols_override` (test harness only, default None = the reference's arithmetic): integer latent symbols to use in
    place of floor(y - mu + 0.5) - lets a test compare everything DOWNSTREAM of the quantiser "given equal indices" when
    the device decided a rounding tie (|frac - .5| < 1e-4, asserted by the caller) the other way."""
    z = hyper_analysis_forward(sd, latents, prefix + "analysis_net.")
    if noise_hyper is None:
        noise_hyper = torch.empty_like(z).uniform_(-0.5, 0.5)
    noisy_z = z + noise_hyper
    _, nz_bpp = estimate_entropy(factorized_likelihood(sd, noisy_z, prefix + "hyperlatent_likelihood."), spatial_shape)
    quant_z = torch.floor(z + 0.5)
    _, qz_bpp = estimate_entropy(factorized_likelihood(sd, quant_z, prefix + "hyperlatent_likelihood."), spatial_shape)
    z_dec = noisy_z if training else quant_z
    means = hyper_synthesis_forward(sd, z_dec, prefix + "synthesis_mu.")
    scales = lower_bound_toward(hyper_synthesis_forward(sd, z_dec, prefix + "synthesis_std."), MIN_SCALE)
    if noise_latent is None:
        noise_latent = torch.empty_like(latents).uniform_(-0.5, 0.5)
    noisy_y = latents + noise_latent
    _, ny_bpp = estimate_entropy(latent_likelihood(noisy_y, means, scales, likelihood_type), spatial_shape)
    sym = torch.floor(latents - means + 0.5) if symbols_override is None else symbols_override.to(latents.dtype)
    quant_y = sym.detach() + means
    _, qy_bpp = estimate_entropy(latent_likelihood(quant_y, means, scales, likelihood_type), spatial_shape)
    # quantize_latents_st (hyperprior.py:108-122)
    vals = latents - means
    decoded = vals + (sym - vals).detach() + means
    return HyperInfo(decoded, ny_bpp, nz_bpp, ny_bpp + nz_bpp, qy_bpp, qz_bpp, qy_bpp + qz_bpp, means, scales, z)


def quantized_indices(latents, means):
    """Integer symbols floor(y - mu + 0.5) (src/hyperprior.py:70-72; what the entropy coder consumes)."""
    return torch.floor(latents - means + 0.5).to(torch.int64)


# ---- LPIPS (alex, net-lin v0.1) -------------------------------------------------------------------------
ALEX_CFG = [(3, 64, 11, 4, 2, True), (64, 192, 5, 1, 2, True), (192, 384, 3, 1, 1, False),
            (384, 256, 3, 1, 1, False), (256, 256, 3, 1, 1, False)]
ALEX_IDX = [0, 3, 6, 8, 10]
LPIPS_SHIFT = (-.030, -.088, -.188)
LPIPS_SCALE = (.458, .448, .450)


def alexnet_taps(backbone, x):
    """torchvision AlexNet.features sliced at relu1..relu5 (pretrained_networks.py:59-94)."""
    taps = []
    h = x
    for li, (idx, (ci, co, k, s, p, pool)) in enumerate(zip(ALEX_IDX, ALEX_CFG)):
        h = F.relu(F.conv2d(h, backbone[f"features.{idx}.weight"], backbone[f"features.{idx}.bias"], stride=s,
                            padding=p))
        taps.append(h)
        if pool and li < 4:
            h = F.max_pool2d(h, kernel_size=3, stride=2)
    return taps


# torchvision VGG16.features (pretrained_networks.py:96-134): conv index in the Sequential -> (C_in, C_out), 2x2 max pool
# BEFORE the convs at 5, 10, 17, 24; LPIPS taps relu1_2, relu2_2, relu3_3, relu4_3, relu5_3 = the ReLUs after convs 2, 7,
# 14, 21, 28
VGG_CONVS = [(0, 3, 64), (2, 64, 64), (5, 64, 128), (7, 128, 128), (10, 128, 256), (12, 256, 256), (14, 256, 256),
             (17, 256, 512), (19, 512, 512), (21, 512, 512), (24, 512, 512), (26, 512, 512), (28, 512, 512)]
VGG_POOL_BEFORE = (5, 10, 17, 24)
VGG_TAPS = (2, 7, 14, 21, 28)


def vgg16_taps(backbone, x):
    taps = []
    h = x
    for idx, ci, co in VGG_CONVS:
        if idx in VGG_POOL_BEFORE:
            h = F.max_pool2d(h, kernel_size=2, stride=2)
        h = F.relu(F.conv2d(h, backbone[f"features.{idx}.weight"], backbone[f"features.{idx}.bias"], stride=1, padding=1))
        if idx in VGG_TAPS:
            taps.append(h)
    return taps


def lpips_forward(backbone, lins, pred, target, normalize=True, net="alex"):
    """perceptual_loss.py:26-46 + networks_basic.py:61-108: returns (N,1,1,1).  in0 = target, in1 = pred."""
    if normalize:
        target = 2 * target - 1
        pred = 2 * pred - 1
    shift = torch.tensor(LPIPS_SHIFT, dtype=pred.dtype).view(1, 3, 1, 1)
    scale = torch.tensor(LPIPS_SCALE, dtype=pred.dtype).view(1, 3, 1, 1)
    taps_fn = alexnet_taps if net == "alex" else vgg16_taps
    t0 = taps_fn(backbone, (target - shift) / scale)
    t1 = taps_fn(backbone, (pred - shift) / scale)
    val = 0
    for k in range(5):
        f0 = t0[k] / torch.sqrt(torch.sum(t0[k] ** 2, dim=1, keepdim=True) + 1e-10)
        f1 = t1[k] / torch.sqrt(torch.sum(t1[k] ** 2, dim=1, keepdim=True) + 1e-10)
        d = (f0 - f1) ** 2
        w = lins[k].view(1, -1, 1, 1).to(d.dtype)
        val = val + (d * w).sum(dim=1, keepdim=True).mean(dim=[2, 3], keepdim=True)
    return val


# ---- Discriminator --------------------------------------------------------------------------------------
def spectral_norm_weight(w_orig, u, v, training, eps=1e-12):
    """torch.nn.utils.spectral_norm (legacy hook) used at discriminator.py:46-62: returns (weight, u', v')."""
    wm = w_orig.reshape(w_orig.shape[0], -1)
    u, v = u.clone(), v.clone()
    if training:
        with torch.no_grad():
            v = F.normalize(torch.mv(wm.t(), u), dim=0, eps=eps)
            u = F.normalize(torch.mv(wm, v), dim=0, eps=eps)
    sigma = torch.dot(u, torch.mv(wm, v))
    return w_orig / sigma, u, v


def discriminator_forward(sd, x, y, training=True, prefix="Discriminator."):
    """src/network/discriminator.py:66-86.  Returns (sigmoid, logits, new_uv dict)."""
    p = prefix
    c = F.leaky_relu(F.conv2d(reflect_pad(y, 1, 1, 1, 1), sd[p + "context_conv.weight"], sd[p + "context_conv.bias"]),
                     0.2)
    c = F.interpolate(c, scale_factor=16, mode="nearest")
    h = torch.cat((x, c), dim=1)
    new_uv = {}
    for i in range(1, 5):
        w, u, v = spectral_norm_weight(sd[p + f"conv{i}.weight_orig"], sd[p + f"conv{i}.weight_u"],
                                       sd[p + f"conv{i}.weight_v"], training)
        new_uv[p + f"conv{i}.weight_u"], new_uv[p + f"conv{i}.weight_v"] = u, v
        h = F.leaky_relu(F.conv2d(reflect_pad(h, 1, 1, 1, 1), w, sd[p + f"conv{i}.bias"], stride=2), 0.2)
    logits = F.conv2d(h, sd[p + "conv_out.weight"], sd[p + "conv_out.bias"]).view(-1, 1)
    return torch.sigmoid(logits), logits, new_uv


# ---- losses / model ---------------------------------------------------------------------------------------
def get_scheduled_params(param, schedule, step_counter, ignore_schedule=False):
    """src/helpers/utils.py:64-72."""
    if not ignore_schedule:
        vals, steps = schedule["vals"], schedule["steps"]
        idx = int(np.where(step_counter < np.array(steps + [step_counter + 1]))[0][0])
        param = param * vals[idx]
    return param


DEFAULT_ARGS = dict(k_M=0.075 * 2 ** (-5), k_P=1., beta=0.15, lambda_B=2 ** (-4), lambda_A=2, target_rate=0.14,
                    lambda_schedule=dict(vals=[2., 1.], steps=[50000]),
                    target_schedule=dict(vals=[0.20 / 0.14, 1.], steps=[50000]), ignore_schedule=False)


def model_forward(sd, backbone, lins, x, step_counter=1, training=True, gan=False, train_generator=True,
                  noise_hyper=None, noise_latent=None, args=None, n_residual_blocks=9, symbols_override=None):
    """src/model.py:346-387 (TRAINING/VALIDATION modes): returns dict(losses..., intermediates...)."""
    a = dict(DEFAULT_ARGS)
    if args:
        a.update(args)
    y = encoder_forward(sd, x)
    hi = hyperprior_forward(sd, y, x.shape[2:], training, noise_hyper, noise_latent, symbols_override=symbols_override)
    x_gen = generator_forward(sd, hi.decoded, n_residual_blocks)
    x_l, xg_l = x, x_gen
    if a.get("normalize_input_image", False):                                     # model.py:155-156, 206-209
        x_gen = torch.tanh(x_gen)
        x_l, xg_l = (x + 1.) / 2., (x_gen + 1.) / 2.
    distortion = torch.mean((xg_l * 255. - x_l * 255.) ** 2)                      # model.py:190-194
    perceptual = torch.mean(lpips_forward(backbone, lins, xg_l, x_l, normalize=True))   # model.py:196-199
    lam_A = get_scheduled_params(a["lambda_A"], a["lambda_schedule"], step_counter, a["ignore_schedule"])
    lam_B = get_scheduled_params(a["lambda_B"], a["lambda_schedule"], step_counter, a["ignore_schedule"])
    target = get_scheduled_params(a["target_rate"], a["target_schedule"], step_counter, a["ignore_schedule"])
    rate_penalty = lam_A if hi.total_qbpp.item() > target else lam_B              # losses.py:21-25
    compression = rate_penalty * hi.total_nbpp + a["k_M"] * distortion + a["k_P"] * perceptual
    out = dict(y=y, hyperinfo=hi, reconstruction=x_gen, distortion=distortion, perceptual=perceptual,
               rate_penalty=rate_penalty)
    if gan:
        xg = x_gen if train_generator else x_gen.detach()
        d_in = torch.cat([x, xg], dim=0)                                          # model.py:176
        lat = torch.repeat_interleave(hi.decoded.detach(), 2, dim=0)              # model.py:178-179 (pairing quirk)
        d_out, d_logits, new_uv = discriminator_forward(sd, d_in, lat, training)
        d_logits = d_logits.squeeze()
        real_logits, gen_logits = torch.chunk(d_logits, 2, dim=0)
        bce = F.binary_cross_entropy_with_logits
        d_loss = bce(real_logits, torch.ones_like(real_logits)) + bce(gen_logits, torch.zeros_like(gen_logits))
        g_loss = bce(gen_logits, torch.ones_like(gen_logits))                     # losses.py:30-41
        compression = compression + a["beta"] * g_loss                            # model.py:373-375
        out.update(disc=d_loss, g_loss=g_loss, new_uv=new_uv, d_logits=d_logits)
    out["compression"] = compression
    return out


# ---- deterministic fixtures ---------------------------------------------------------------------------------
def _shapes(C=220, N=320, n_res=9, gan=True):
    """(key, shape, kind) for every tensor of the reference Model.state_dict() [probe: 168 tensors with GAN]."""
    s = []
    enc = [(3, 60, 7), (60, 120, 3), (120, 240, 3), (240, 480, 3), (480, 960, 3)]
    for i, (ci, co, k) in enumerate(enc, 1):
        s += [(f"Encoder.conv_block{i}.1.weight", (co, ci, k, k), "w"), (f"Encoder.conv_block{i}.1.bias", (co,), "b"),
              (f"Encoder.conv_block{i}.2.gamma", (1, co, 1, 1), "g"), (f"Encoder.conv_block{i}.2.beta", (1, co, 1, 1), "be")]
    s += [("Encoder.conv_block_out.1.weight", (C, 960, 3, 3), "w"), ("Encoder.conv_block_out.1.bias", (C,), "b")]
    s += [("Generator.conv_block_init.0.gamma", (1, C, 1, 1), "g"), ("Generator.conv_block_init.0.beta", (1, C, 1, 1), "be"),
          ("Generator.conv_block_init.2.weight", (960, C, 3, 3), "w"), ("Generator.conv_block_init.2.bias", (960,), "b"),
          ("Generator.conv_block_init.3.gamma", (1, 960, 1, 1), "g"), ("Generator.conv_block_init.3.beta", (1, 960, 1, 1), "be")]
    for m in range(n_res):
        for c in ("conv1", "conv2"):
            s += [(f"Generator.resblock_{m}.{c}.weight", (960, 960, 3, 3), "w"), (f"Generator.resblock_{m}.{c}.bias", (960,), "b")]
        for n in ("norm1", "norm2"):
            s += [(f"Generator.resblock_{m}.{n}.gamma", (1, 960, 1, 1), "g"), (f"Generator.resblock_{m}.{n}.beta", (1, 960, 1, 1), "be")]
    up = [(960, 480), (480, 240), (240, 120), (120, 60)]
    for i, (ci, co) in enumerate(up, 1):
        s += [(f"Generator.upconv_block{i}.0.weight", (ci, co, 3, 3), "wt"), (f"Generator.upconv_block{i}.0.bias", (co,), "b"),
              (f"Generator.upconv_block{i}.1.gamma", (1, co, 1, 1), "g"), (f"Generator.upconv_block{i}.1.beta", (1, co, 1, 1), "be")]
    s += [("Generator.conv_block_out.1.weight", (3, 60, 7, 7), "w"), ("Generator.conv_block_out.1.bias", (3,), "b")]
    s += [("Hyperprior.analysis_net.conv1.weight", (N, C, 3, 3), "w"), ("Hyperprior.analysis_net.conv1.bias", (N,), "b"),
          ("Hyperprior.analysis_net.conv2.weight", (N, N, 5, 5), "w"), ("Hyperprior.analysis_net.conv2.bias", (N,), "b"),
          ("Hyperprior.analysis_net.conv3.weight", (N, N, 5, 5), "w"), ("Hyperprior.analysis_net.conv3.bias", (N,), "b")]
    for nm in ("synthesis_mu", "synthesis_std"):
        s += [(f"Hyperprior.{nm}.conv1.weight", (N, N, 5, 5), "wt"), (f"Hyperprior.{nm}.conv1.bias", (N,), "b"),
              (f"Hyperprior.{nm}.conv2.weight", (N, N, 5, 5), "wt"), (f"Hyperprior.{nm}.conv2.bias", (N,), "b"),
              (f"Hyperprior.{nm}.conv3.weight", (N, C, 3, 3), "wt"), (f"Hyperprior.{nm}.conv3.bias", (C,), "b")]
    f = (1, 3, 3, 3, 1)
    for k in range(4):
        s += [(f"Hyperprior.hyperlatent_likelihood.H_{k}", (N, f[k + 1], f[k]), "H"),
              (f"Hyperprior.hyperlatent_likelihood.a_{k}", (N, f[k + 1], 1), "a"),
              (f"Hyperprior.hyperlatent_likelihood.b_{k}", (N, f[k + 1], 1), "bb")]
    if gan:
        s += [("Discriminator.context_conv.weight", (12, C, 3, 3), "w"), ("Discriminator.context_conv.bias", (12,), "b")]
        dch = [(15, 64), (64, 128), (128, 256), (256, 512)]
        for i, (ci, co) in enumerate(dch, 1):
            s += [(f"Discriminator.conv{i}.bias", (co,), "b"), (f"Discriminator.conv{i}.weight_orig", (co, ci, 4, 4), "w"),
                  (f"Discriminator.conv{i}.weight_u", (co,), "u"), (f"Discriminator.conv{i}.weight_v", (ci * 16,), "u")]
        s += [("Discriminator.conv_out.weight", (1, 512, 1, 1), "w"), ("Discriminator.conv_out.bias", (1,), "b")]
    return s


def make_state_dict(seed=0, C=220, N=320, n_res=9, gan=True, dtype=torch.float32):
    """Seeded, construction-order-independent weights: fan-in-scaled uniform conv weights, non-trivial gamma/beta/
    biases (so affine terms are exercised), reference-like factorised-prior init perturbed, unit-norm u/v."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    scale_H = 10. ** 0.25
    f = (1, 3, 3, 3, 1)
    for key, shape, kind in _shapes(C, N, n_res, gan):
        r = torch.rand(shape, generator=g, dtype=torch.float64)
        if kind in ("w", "wt"):
            fan_in = (shape[1] if kind == "w" else shape[0]) * shape[2] * shape[3]
            t = (r * 2 - 1) * math.sqrt(3.0 / fan_in)
        elif kind == "b":
            t = (r * 2 - 1) * 0.1
        elif kind == "g":
            t = 0.75 + 0.5 * r
        elif kind == "be":
            t = (r * 2 - 1) * 0.1
        elif kind == "H":
            k = int(key[-1])
            t = math.log(math.expm1(1 / scale_H / f[k + 1])) + (r * 2 - 1) * 0.2
        elif kind == "a":
            t = (r * 2 - 1) * 0.3
        elif kind == "bb":
            t = (r * 2 - 1) * 0.5
        elif kind == "u":
            t = r * 2 - 1
            t = t / t.norm()
        sd[key] = t.to(dtype)
    return sd


def make_alex_backbone(seed=1234, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for idx, (ci, co, k, s, p, _) in zip(ALEX_IDX, ALEX_CFG):
        bound = 1.0 / math.sqrt(ci * k * k)
        sd[f"features.{idx}.weight"] = ((torch.rand((co, ci, k, k), generator=g) * 2 - 1) * bound).to(dtype)
        sd[f"features.{idx}.bias"] = ((torch.rand((co,), generator=g) * 2 - 1) * bound).to(dtype)
    return sd


def make_vgg_backbone(seed=4321, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for idx, ci, co in VGG_CONVS:
        bound = math.sqrt(6.0 / (ci * 9))                 # keeps activations alive through 13 ReLU layers
        sd[f"features.{idx}.weight"] = ((torch.rand((co, ci, 3, 3), generator=g) * 2 - 1) * bound).to(dtype)
        sd[f"features.{idx}.bias"] = ((torch.rand((co,), generator=g) * 2 - 1) * 0.05).to(dtype)
    return sd


def make_image(seed, B, H=256, W=256, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return torch.rand((B, 3, H, W), generator=g).to(dtype)


def make_noise(seed, shape, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) - 0.5).to(dtype)


# ---- EVALUATION path: device half of `compress` ------------------------------------------------
PRIOR_SCALES_MIN, PRIOR_SCALES_MAX, PRIOR_SCALES_LEVELS = 0.11, 256, 64          # prior_model.py:19-21


def prior_scale_table(scales_min=PRIOR_SCALES_MIN, scales_max=PRIOR_SCALES_MAX, levels=PRIOR_SCALES_LEVELS):
    """prior_model.py:23-25"""
    import numpy as np
    return torch.Tensor(np.exp(np.linspace(np.log(scales_min), np.log(scales_max), levels)))


def prior_compute_indices(scales, scale_table, scales_min=PRIOR_SCALES_MIN):
    """prior_model.py:148-156: index of the table entry used for each predicted scale (int32)."""
    scales = torch.clamp(scales, min=scales_min)                       # LowerBoundToward forward, maths.py:87-95
    indices = torch.ones_like(scales, dtype=torch.int32) * (len(scale_table) - 1)
    for s in scale_table[:-1]:
        indices = indices - (scales <= s).to(torch.int32)
    return indices


def prior_symbols(latents, means):
    """prior_model.py:180-181"""
    return torch.floor(latents + 0.5 - means).to(torch.int32)


def hyper_symbols_and_indices(hyperlatents):
    """hyperprior_model.py:135-139,160-169"""
    N, C, H, W = hyperlatents.shape
    indices = torch.arange(C, dtype=torch.int32).view(-1, 1, 1).repeat(1, H, W)
    indices = torch.repeat_interleave(indices.unsqueeze(0), repeats=N, dim=0)
    return torch.floor(hyperlatents + 0.5).to(torch.int32), indices


def make_symbol_inputs(seed=11, shape=(2, 6, 5, 7)):
    """Latents with exact half-integer residuals (rounding ties), scales on / around every table entry."""
    g = torch.Generator().manual_seed(seed)
    table = prior_scale_table()
    means = torch.randn(shape, generator=g) * 3
    lat = means + torch.randn(shape, generator=g) * 4
    lat.view(-1)[::7] = (means.view(-1)[::7] + torch.randint(-6, 6, (means.view(-1)[::7].numel(),), generator=g).float()
                         + 0.5)
    scales = torch.exp(torch.randn(shape, generator=g) * 2.5)
    flat = scales.view(-1)
    n = min(flat.numel() // 3, len(table))
    flat[:n] = table[:n]                                        # exactly on a table entry (the `<=` boundary)
    flat[n:2 * n] = table[:n] * (1 + 1e-6)
    flat[2 * n:3 * n] = table[:n] * (1 - 1e-6)
    flat[-3:] = torch.tensor([0.0, 0.05, 1e4])                  # below SCALES_MIN, above the table
    return lat, means, scales, table

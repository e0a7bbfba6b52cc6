"""The backward pass pinned at the sizes that are benchmarked.

1. BASELINE configs[2] shape - compression_gan, batch 16 x 256 x 256, 9 residual blocks, float32 parity mode - one G-turn
   and one D-turn as the reference's loop runs them (train.py:119-141; src/model.py:346-387): EVERY parameter gradient
   (amortisation nets, hyperprior density, Discriminator incl. the gradients the G-turn leaves behind for the D step,
   SURVEY section 3.2) against the oracle's autograd, written through the ParamArena slots exactly as the benchmark does.
   Bar: max |g - g_ref| <= 1e-3 x max |g_ref| per tensor.  Where the device decided a rounding tie of the latent
   quantiser the other way (asserted: |frac - .5| < 1e-4, <= 1e-4 of the indices) the oracle is evaluated "given equal
   indices" (oracle `symbols_override`), because one flipped index moves a decoded latent by 1.0.
2. BASELINE configs[4] per-GPU shape - regime high, ONE 1024 x 1024 crop (default_config.py:82-86): float32 forward +
   backward against the oracle, bf16 finite and bit-reproducible.
The planner chooses kernels by grid size, so these are different plans from the 128 x 128 module tests."""
import os

import numpy as np
import pytest
import torch

from oracle import hific_oracle as O
from gradcheck import check_grads, relu_trace

pytestmark = [pytest.mark.gpu, pytest.mark.slow]

HERE = os.path.dirname(os.path.abspath(__file__))


def _lins():
    w = np.load(os.path.join(os.path.dirname(HERE), "high-fidelity-generative-compression_amd", "loss", "weights",
                             "lpips_alex_lin_v0.1.npz"))
    return [torch.from_numpy(w[f"lin{i}"].copy()) for i in range(5)]


def _build(hific, dev, dt, B, S, regime="low", gan=True):
    import hific_amd
    from hific_amd import optim
    from hific_amd.default_config import make_args, hific_args, mse_lpips_args, ModelTypes
    hific.set_compute_dtype(dt)
    args = make_args(hific_args if gan else mse_lpips_args, regime=regime, batch_size=B, image_dims=(3, S, S),
                     latent_dims=(220, S // 16, S // 16))
    model = hific_amd.Model(args, model_type=ModelTypes.COMPRESSION_GAN if gan else ModelTypes.COMPRESSION,
                            allow_random_lpips_backbone=True)
    sd = O.make_state_dict(seed=0, gan=gan)
    model.load_state_dict(sd, strict=True)
    model.perceptual_loss.load_backbone_state_dict(O.make_alex_backbone())
    model = model.to(dev).train()
    model.Hyperprior.keep_debug = True
    # the three optimizer groups of train.py:287-301 as arenas: backward kernels write the gradient slots
    amort = [p for m in model.amortization_models for p in m.parameters()]
    hyper = list(model.Hyperprior.hyperlatent_likelihood.parameters())
    arenas = {"amort": optim.ParamArena(amort), "hyper": optim.ParamArena(hyper)}
    if gan:
        arenas["disc"] = optim.ParamArena(list(model.Discriminator.parameters()))
    return model, sd, arenas, args


def _device_symbols(model, inter):
    mu = model.Hyperprior.debug_latent_means
    return torch.round(inter.latents_quantized.detach().float() - mu).cpu()


def _oracle_turn(sdr, x, nh, nl, train_generator, sym_dev, oargs=None, dt=torch.float32):
    """Oracle forward for one turn in dtype `dt`, given the device's symbols wherever they differ from the oracle's own -
    which is only allowed at rounding ties (asserted against the float32 oracle, the reference's arithmetic)."""
    bb, lins = O.make_alex_backbone(dtype=dt), [l.to(dt) for l in _lins()]
    x, nh, nl = x.to(dt), nh.to(dt), nl.to(dt)
    kw = dict(step_counter=1, training=True, gan=True, train_generator=train_generator, noise_hyper=nh, noise_latent=nl,
              args=oargs)
    with torch.no_grad():
        y = O.encoder_forward(sdr, x)
        hi = O.hyperprior_forward(sdr, y, x.shape[2:], True, nh, nl)
    sym_o = torch.floor(y - hi.latent_means + 0.5)
    flips = sym_o != sym_dev.to(dt)
    n = int(flips.sum())
    override = None
    if n:
        frac = y - hi.latent_means + 0.5
        frac = frac - torch.floor(frac)
        tie = torch.minimum(frac, 1 - frac)
        print(f"  rounding-tie flips vs the {dt} oracle: {n} of {flips.numel()}, max tie distance {float(tie[flips].max()):.2e}")
        assert n <= max(2, 1e-4 * flips.numel()) and float(tie[flips].max()) < 1e-4
        assert float((sym_o - sym_dev.to(dt)).abs().max()) <= 1
        override = sym_dev
    return O.model_forward(sdr, bb, lins, x, symbols_override=override, **kw)


def _params_for_autograd(sd, names, dt):
    return {k: (v.to(dt).clone().requires_grad_(True) if k in names else (v.to(dt) if v.dtype.is_floating_point else v.clone()))
            for k, v in sd.items()}


def test_gan_cycle_every_gradient_fullsize_f32(hific, dev):
    B, S = 16, 256
    model, sd, arenas, _ = _build(hific, dev, torch.float32, B, S)
    xs = [O.make_image(21, B, S, S), O.make_image(22, B, S, S)]
    nz = [(O.make_noise(31, (B, 320, S // 64, S // 64)), O.make_noise(32, (B, 220, S // 16, S // 16))),
          (O.make_noise(33, (B, 320, S // 64, S // 64)), O.make_noise(34, (B, 220, S // 16, S // 16)))]
    noises = [t.to(dev) for pair in nz for t in pair]
    model.Hyperprior._draw_noise = lambda t: noises.pop(0)
    params = dict(model.named_parameters())
    disc_keys = [k for k in params if k.startswith("Discriminator.")]
    gen_keys = [k for k in params if not k.startswith("Discriminator.")]

    # ---- device: G-turn, then D-turn on the next batch; the Discriminator slots are NOT cleared in between ----------
    losses, inter = model(xs[0].to(dev), train_generator=True, return_intermediates=True, writeout=False)
    losses["compression"].backward()
    torch.cuda.synchronize()
    dev_G = {k: params[k].grad.detach().float().cpu().clone() for k in params}
    loss_G, sym_G = float(losses["compression"]), _device_symbols(model, inter)
    assert all(not s.fresh for a in arenas.values() for s in a.slots), "a parameter received no gradient on the G-turn"
    arenas["amort"].zero_grad(); arenas["hyper"].zero_grad()
    losses, inter = model(xs[1].to(dev), train_generator=False, return_intermediates=True, writeout=False)
    losses["disc"].backward()
    torch.cuda.synchronize()
    dev_D = {k: params[k].grad.detach().float().cpu().clone() for k in disc_keys}
    loss_D, sym_D = float(losses["disc"]), _device_symbols(model, inter)
    # a D-turn sends nothing into the Encoder / Generator / hyper nets (x_gen and the latents are detached, model.py:171-179)
    assert all(s.fresh for s in arenas["amort"].slots) and all(s.fresh for s in arenas["hyper"].slots)
    uv_dev = {k: v.detach().cpu().clone() for k, v in model.state_dict().items() if "weight_u" in k or "weight_v" in k}
    del model, arenas, losses, inter
    torch.cuda.empty_cache()

    # ---- oracle: the same two turns with torch autograd on the CPU (float32 = the bar; float64 = the arbiter) ---------
    def oracle_cycle(dt):
        sdr = _params_for_autograd(sd, params, dt)
        out = _oracle_turn(sdr, xs[0], *nz[0], True, sym_G, dt=dt)
        out["compression"].backward()
        res = dict(loss_G=float(out["compression"]), G={k: sdr[k].grad.detach().clone() for k in params})
        for k in gen_keys:
            sdr[k].grad = None
        for k, v in out["new_uv"].items():         # spectral-norm power iteration state carries over (in-place buffers)
            sdr[k] = v.detach()
        out = _oracle_turn(sdr, xs[1], *nz[1], False, sym_D, dt=dt)
        out["disc"].backward()
        assert all(sdr[k].grad is None for k in gen_keys)
        res.update(loss_D=float(out["disc"]), D={k: sdr[k].grad.detach().clone() for k in disc_keys},
                   uv={k: v.detach() for k, v in out["new_uv"].items()})
        return res

    with relu_trace(O) as trace:                     # pre-activations of every ReLU / LeakyReLU of the float32 oracle run
        o32 = oracle_cycle(torch.float32)
    assert abs(loss_G - o32["loss_G"]) < 1e-3 * abs(o32["loss_G"]) and abs(loss_D - o32["loss_D"]) < 1e-3 * abs(o32["loss_D"])
    cache = {}

    def o64():
        if not cache:
            cache.update(oracle_cycle(torch.float64))
        return cache

    worst_G, arb_G = check_grads(dev_G, o32["G"], lambda: o64()["G"], 1e-3, "G-turn, all parameters", trace=trace)
    worst_D, arb_D = check_grads(dev_D, o32["D"], lambda: o64()["D"], 1e-3, "D-turn, Discriminator (G-turn leftovers + D-turn)",
                                 trace=trace)
    for k, v in o32["uv"].items():
        assert torch.allclose(uv_dev[k], v, atol=1e-5), k
    print(f"  full-size f32 gradients vs oracle: worst G-turn {worst_G:.2e}, worst D-turn {worst_D:.2e}; beyond the plain "
          f"1e-3 bar and judged separately (tests/gradcheck.py): {sorted(arb_G) + sorted(arb_D)}")
    assert len(arb_G) <= 8 and len(arb_D) <= 2, "too many tensors needed a second opinion"


def test_config5_one_1024_crop_regime_high(hific, dev):
    """BASELINE configs[4] on one GPU: 1 x 3 x 1024 x 1024, regime high (target 0.45 bpp, lambda_A 0.5)."""
    B, S = 1, 1024
    oargs = dict(lambda_A=2 ** (-1), target_rate=0.45)
    x = O.make_image(41, B, S, S)
    nh, nl = O.make_noise(42, (B, 320, S // 64, S // 64)), O.make_noise(43, (B, 220, S // 16, S // 16))

    def run(dt):
        model, sd, arenas, args = _build(hific, dev, dt, B, S, regime="high")
        assert args.target_rate == 0.45 and args.lambda_A == 0.5
        noises = [nh.to(dev), nl.to(dev)]
        model.Hyperprior._draw_noise = lambda t: noises.pop(0)
        losses, inter = model(x.to(dev), train_generator=True, return_intermediates=True, writeout=False)
        losses["compression"].backward()
        torch.cuda.synchronize()
        params = dict(model.named_parameters())
        grads = {k: p.grad.detach().float().cpu().clone() for k, p in params.items()}
        res = dict(loss=float(losses["compression"]), disc=float(losses["disc"]), n_bpp=float(inter.n_bpp),
                   q_bpp=float(inter.q_bpp), rec=inter.reconstruction.detach().float().cpu(), grads=grads,
                   sym=_device_symbols(model, inter), sd=sd)
        del model, arenas, losses, inter
        torch.cuda.empty_cache()
        return res

    r32 = run(torch.float32)

    def oracle(dt):
        sdr = _params_for_autograd(r32["sd"], r32["grads"], dt)
        out = _oracle_turn(sdr, x, nh, nl, True, r32["sym"], oargs, dt=dt)
        out["compression"].backward()
        return out, {k: sdr[k].grad.detach() for k in r32["grads"]}

    with relu_trace(O) as trace:
        out, g32 = oracle(torch.float32)
    for name, a, b in (("compression", r32["loss"], float(out["compression"])), ("disc", r32["disc"], float(out["disc"])),
                       ("n_bpp", r32["n_bpp"], float(out["hyperinfo"].total_nbpp)),
                       ("q_bpp", r32["q_bpp"], float(out["hyperinfo"].total_qbpp))):
        assert abs(a - b) < 1e-3 * abs(b), (name, a, b)
    rec_ref = out["reconstruction"].detach()
    assert float((r32["rec"] - rec_ref).abs().max()) < 1e-3 * float(rec_ref.abs().max())
    _, arb = check_grads(r32["grads"], g32, lambda: oracle(torch.float64)[1], 1e-3, "config 5, f32, all parameters",
                         trace=trace)
    assert len(arb) <= 8
    # the benchmarked mode at this shape: finite, bit-reproducible, same rate decision, aggregates near the f32 run
    b1, b2 = run(torch.bfloat16), run(torch.bfloat16)
    assert all(torch.isfinite(g).all() for g in b1["grads"].values()) and np.isfinite(b1["loss"])
    assert b1["loss"] == b2["loss"] and torch.equal(b1["rec"], b2["rec"])
    assert all(torch.equal(b1["grads"][k], b2["grads"][k]) for k in b1["grads"])
    assert abs(b1["loss"] - r32["loss"]) < 1e-2 * abs(r32["loss"]) and abs(b1["q_bpp"] - r32["q_bpp"]) < 3e-3 * r32["q_bpp"]
    n_f = int((b1["sym"] != r32["sym"]).sum())
    print(f"  config 5 bf16 (exact-index chain) vs f32 device: loss rel {abs(b1['loss'] - r32['loss']) / abs(r32['loss']):.2e}, "
          f"index flips {n_f} of {b1['sym'].numel()}")
    assert n_f <= max(2, 1e-4 * b1["sym"].numel())


# Fixed ceilings (committed by hand, round 6) for a bf16-operand backward pass against the float32 ORACLE at 16 x 256^2:
# max |g - g_oracle| / max |g_oracle| per tensor, by parameter class = ~2x the worst value measured on the MI355X over both
# bf16 modes (exact training | plain: Encoder 5.4e-3 | 9.5e-3, Generator head 8.4e-3 | 1.27e-2, residual blocks 7.8e-3 |
# 1.51e-2, up-convolutions 4.0e-3 | 8.2e-3, output conv 1.2e-5 | 1.9e-3, Hyperprior 1.33e-2 | 1.33e-2, Discriminator
# leftovers 6.6e-3 | 8.9e-3; gpurun_out/r06_run2).  A sign error or a dropped term gives 1-2, a wrong saved activation in
# one layer >= 0.3.
BF16_VS_ORACLE_CEILING = {
    "Encoder.": 2e-2, "Generator.conv_block_init": 2.5e-2, "Generator.resblock": 3e-2, "Generator.upconv": 2e-2,
    "Generator.conv_block_out": 5e-3, "Hyperprior.": 3e-2, "Discriminator.": 2e-2,
}


def _ceiling(name):
    for k, v in BF16_VS_ORACLE_CEILING.items():
        if name.startswith(k):
            return k, v
    raise KeyError(name)


def test_bf16_modes_every_G_turn_gradient_against_the_oracle_fullsize(hific, dev):
    """the exact-TRAINING mode (ops.set_exact_training: float32-accurate forward values, bf16
    backward) and the plain bf16 mode, one G-turn at BASELINE configs[2]'s shape each, EVERY parameter gradient (Encoder,
    Generator, Hyperprior incl. the density, and the Discriminator gradients the G-turn leaves behind) element-wise against
    the float32 ORACLE's autograd under fixed per-class ceilings - not against another device mode, not a regenerated table.
    The exact-training forward must also meet north_star's 1e-3 on loss and reconstruction."""
    from hific_amd import ops
    B, S = 16, 256
    x = O.make_image(21, B, S, S)
    nh, nl = O.make_noise(31, (B, 320, S // 64, S // 64)), O.make_noise(32, (B, 220, S // 16, S // 16))
    res = {}
    for mode in ("exact_training", "plain"):
        ops.set_exact_training(mode == "exact_training")
        try:
            model, sd, arenas, _ = _build(hific, dev, torch.bfloat16, B, S)
            noises = [nh.to(dev), nl.to(dev)]
            model.Hyperprior._draw_noise = lambda t: noises.pop(0)
            params = dict(model.named_parameters())
            losses, inter = model(x.to(dev), train_generator=True, return_intermediates=True, writeout=False)
            losses["compression"].backward()
            torch.cuda.synchronize()
        finally:
            ops.set_exact_training(False)
        res[mode] = dict(grads={k: params[k].grad.detach().float().cpu().clone() for k in params},
                         loss=float(losses["compression"].detach()), sym=_device_symbols(model, inter),
                         rec=inter.reconstruction.detach().float().cpu())
        assert all(not s.fresh for a in arenas.values() for s in a.slots), "a parameter received no gradient on the G-turn"
        del model, arenas, losses, inter
        torch.cuda.empty_cache()
    hific.set_compute_dtype(torch.float32)
    names = list(res["plain"]["grads"])
    oracle = {}

    def oracle_for(sym):
        key = sym.numpy().tobytes()
        if key not in oracle:
            sdr = _params_for_autograd(sd, names, torch.float32)
            out = _oracle_turn(sdr, x, nh, nl, True, sym)
            out["compression"].backward()
            oracle[key] = (float(out["compression"]), out["reconstruction"].detach(), {k: sdr[k].grad.detach() for k in names})
        return oracle[key]

    report = {}
    for mode, r in res.items():
        loss_o, rec_o, g_o = oracle_for(r["sym"])
        worst = {}
        for k in names:
            g = r["grads"][k]
            assert torch.isfinite(g).all(), (mode, k)
            e = float((g.double() - g_o[k].double()).abs().max()) / max(float(g_o[k].abs().max()), 1e-30)
            cls, cap = _ceiling(k)
            if e > worst.get(cls, (0.0, ""))[0]:
                worst[cls] = (e, k)
        rec_rel = float((r["rec"] - rec_o).abs().max()) / float(rec_o.abs().max())
        loss_rel = abs(r["loss"] - loss_o) / abs(loss_o)
        report[mode] = (worst, rec_rel, loss_rel)
        print(f"  [{mode}] vs the float32 oracle: loss rel {loss_rel:.2e}, reconstruction max-rel {rec_rel:.2e}; worst gradient "
              f"error per class:")
        for cls, (e, k) in sorted(worst.items()):
            print(f"    {cls:28s} {e:.3e}  ({k})   ceiling {BF16_VS_ORACLE_CEILING[cls]:.0e}")
    for mode, (worst, rec_rel, loss_rel) in report.items():
        for cls, (e, k) in worst.items():
            assert e <= BF16_VS_ORACLE_CEILING[cls], (mode, k, e)
    _, rec_rel, loss_rel = report["exact_training"]
    assert rec_rel < 1e-3 and loss_rel < 1e-3, (rec_rel, loss_rel)

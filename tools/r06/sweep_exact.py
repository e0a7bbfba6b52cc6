"""Per-layer sensitivity of the exact Generator chain: which layer groups need split-bf16 operands,
and does a two-product contraction (split only the activation, or only the weight) hold north_star's 1e-3 on the reconstruction?

    python tools/r06/sweep_exact.py            (GPU box; the oracle Generator runs once on the host cores)

The chain always runs the three-product kernels; a degraded layer is emulated exactly by zeroing the `lo` half of an operand
image: activation lo = 0 is what a bf16-stored activation gives the contraction, weight lo = 0 is what a bf16-packed weight
gives it (x*w ~ xh*wh + xl*wh + xh*wl with the dropped term's operand zero).  Reported: reconstruction max-rel error against
the oracle Generator on the same decoded latents, batch 16 x 256^2, 9 residual blocks."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import hific_amd  # noqa: E402
from hific_amd import ops  # noqa: E402
from hific_amd.default_config import make_args, hific_args, ModelTypes  # noqa: E402
from oracle import hific_oracle as O  # noqa: E402

B, S = 16, 256
dev = torch.device("cuda:0")
hific_amd.set_compute_dtype(torch.bfloat16)
args = make_args(hific_args, batch_size=B, image_dims=(3, S, S), latent_dims=(220, S // 16, S // 16))
model = hific_amd.Model(args, model_type=ModelTypes.COMPRESSION_GAN, allow_random_lpips_backbone=True)
sd = O.make_state_dict(seed=0, gan=True)
model.load_state_dict(sd, strict=True)
model.perceptual_loss.load_backbone_state_dict(O.make_alex_backbone())
model = model.to(dev).train()
x = O.make_image(21, B, S, S).to(dev)
noises = [O.make_noise(31, (B, 320, S // 64, S // 64)).to(dev), O.make_noise(32, (B, 220, S // 16, S // 16)).to(dev)]
model.Hyperprior._draw_noise = lambda t: noises.pop(0)
ops.set_exact_training(True)
with torch.no_grad():
    losses, inter = model(x, train_generator=True, return_intermediates=True, writeout=False)
lat = inter.latents_quantized.detach().float()
with torch.no_grad():
    ref = O.generator_forward(sd, lat.cpu(), 9)
G = model.Generator

# ---- layer groups: weight tensor id -> group name -------------------------------------------------------------------
groups = {"head": [G.conv_block_init[2]], "out": [G.conv_block_out[1]]}
for m in range(9):
    blk = getattr(G, f"resblock_{m}")
    groups[f"rb{m}"] = [blk.conv1, blk.conv2]
for i in range(4):
    groups[f"up{i + 1}"] = [getattr(G, f"upconv_block{i + 1}")[0]]
wid = {id(c.weight): g for g, cs in groups.items() for c in cs}
degrade = {}          # group -> (zero activation lo, zero weight lo)


def zero_lo(t, dim, layout, C):
    """Zeroes the lo half of a split image along `dim` (returns a modified clone)."""
    t = t.clone()
    idx = [slice(None)] * t.dim()
    if layout == ops.SPLIT_PAIR:
        for g in range(t.shape[dim] // 32):
            idx[dim] = slice(32 * g + 16, 32 * g + 32)
            t[tuple(idx)] = 0
    else:
        idx[dim] = slice(C, 2 * C) if dim == 1 and t.dtype == torch.bfloat16 else slice(2 * C, 3 * C)
        t[tuple(idx)] = 0
    return t


_get = ops.split_weights.get
_wclones = {}


def get(weight, transposed, layout=ops.SPLIT_3C):
    w3 = _get(weight, transposed, layout)
    za, zw = degrade.get(wid.get(id(weight)), (False, False))
    if not zw:
        return w3
    key = (id(weight), layout)
    if key not in _wclones:
        C = weight.shape[0] if transposed else weight.shape[1]
        _wclones[key] = zero_lo(w3, 0 if transposed else 1, layout, C)       # weights: (hi, hi, lo) -> third block
    return _wclones[key]


ops.split_weights.get = get


def wrap(fn, widx, x3idx, layidx):
    def inner(*a, **k):
        a = list(a)
        w = a[widx]
        za, zw = degrade.get(wid.get(id(w)), (False, False))
        if za:
            lay = a[layidx] if layidx is not None else ops.SPLIT_3C
            a[x3idx] = zero_lo(a[x3idx], 1, lay, a[0].shape[1])                # activations: (hi, lo, hi) -> second block
        return fn(*a, **k)
    return inner


ops.exact_conv_norm = wrap(ops.exact_conv_norm, 2, 1, 11)
ops.exact_conv_transpose_norm = wrap(ops.exact_conv_transpose_norm, 2, 1, 11)
_conv2d = ops.conv2d


def conv2d(x, weight, bias, **k):
    za, zw = degrade.get(wid.get(id(weight)), (False, False))
    if za and k.get("x3") is not None:
        k["x3"] = zero_lo(k["x3"], 1, ops.SPLIT_3C, x.shape[1])
    return _conv2d(x, weight, bias, **k)


ops.conv2d = conv2d


def run(tag, deg):
    degrade.clear(); degrade.update(deg)
    with torch.no_grad():
        rec = G(lat).float().cpu()
    e = float((rec - ref).abs().max() / ref.abs().max())
    print(f"[sweep] {tag:58s} recon max-rel {e:.3e} {'OK' if e < 1e-3 else ''}", flush=True)
    return e


names = ["head"] + [f"rb{m}" for m in range(9)] + [f"up{i}" for i in range(1, 5)] + ["out"]
run("all layers exact (three products)", {})
run("all layers plain (both lo halves zero)", {g: (True, True) for g in names})
run("two products everywhere: activation split only (w lo = 0)", {g: (False, True) for g in names})
run("two products everywhere: weight split only (x lo = 0)", {g: (True, False) for g in names})
for g in names:
    run(f"all exact except {g} plain", {g: (True, True)})
for g in names:
    run(f"all exact except {g} activation-only", {g: (False, True)})
for g in names:
    run(f"all exact except {g} weight-only", {g: (True, False)})
run("trunk (rb0-8) weight-only, rest exact", {f"rb{m}": (True, False) for m in range(9)})
run("trunk (rb0-8) activation-only, rest exact", {f"rb{m}": (False, True) for m in range(9)})

"""Exact-index mode (ops.set_exact_index): in bf16 compute mode the forward contractions of the
chain behind floor(y - mu + 0.5) (Encoder -> analysis -> synthesis_mu; src/hyperprior.py:68-74,108-122,
src/network/encoder.py:104-111) run with split-bf16 operands.  Checked here: the split kernel bit for bit, every conv /
conv-transpose geometry of that chain against float64 CPU convolutions (forward <= 3e-5 of the output scale, i.e. ~500x
tighter than plain bf16; backward at the bf16 tolerance), the derived (hi, hi, lo) weight images across optimizer steps,
and the quantised indices of the modules against the oracle with the tie-aware equality of the float32 tests."""
import pytest
import torch
import torch.nn.functional as F

from oracle import hific_oracle as O

pytestmark = pytest.mark.gpu

# Encoder / hyper-analysis geometries: (N, C, H, W, K, R, stride, (pt, pl, pb, pr), mode)
CONV_CASES = {
    "E1_7x7_reflect3":   (2, 3, 64, 64, 60, 7, 1, (3, 3, 3, 3), "reflect"),
    "E2_3x3s2_asym":     (2, 60, 64, 64, 120, 3, 2, (1, 0, 0, 1), "reflect"),
    "E3_3x3s2_asym":     (2, 120, 32, 32, 240, 3, 2, (1, 0, 0, 1), "reflect"),
    "E5_3x3s2_asym_480": (2, 480, 32, 32, 960, 3, 2, (1, 0, 0, 1), "reflect"),
    "E6_960_220":        (2, 960, 16, 16, 220, 3, 1, (1, 1, 1, 1), "reflect"),
    "A1_zero":           (2, 220, 16, 16, 320, 3, 1, (1, 1, 1, 1), "zeros"),
    "A2_5x5s2_reflect2": (2, 320, 16, 16, 320, 5, 2, (2, 2, 2, 2), "reflect"),
    "A3_5x5s2_8x8":      (3, 320, 8, 8, 320, 5, 2, (2, 2, 2, 2), "reflect"),
    "odd_sizes":         (1, 5, 13, 17, 7, 3, 1, (1, 1, 1, 1), "reflect"),
}
# synthesis_mu geometries: (N, Ci, H, W, Co, R, stride, pad, outpad)
CONVT_CASES = {
    "S1_5x5s2":   (2, 320, 4, 4, 320, 5, 2, 2, 1),
    "S2_5x5s2_8": (3, 320, 8, 8, 320, 5, 2, 2, 1),
    "S3_3x3s1":   (2, 320, 16, 16, 220, 3, 1, 1, 0),
    "odd":        (1, 7, 5, 6, 9, 3, 2, 1, 1),
}
# backward = the ordinary bf16 kernels on operands that are NOT pre-rounded to bf16 here (float32 activations / master
# weights): 2^-9 per operand, measured up to 3e-2 of the gradient scale on the 220 -> 320 layer
FWD_TOL, BWD_TOL = 3e-5, 6e-2


def _rnd(shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(shape, generator=g) * 2 - 1


def _relerr(a, b):
    return (a.double() - b.double()).abs().max().item() / max(b.abs().max().item(), 1e-20)


@pytest.fixture(autouse=True)
def _bf16_exact(hific):
    from hific_amd import ops
    hific.set_compute_dtype(torch.bfloat16)
    was = ops.exact_index_on()
    ops.set_exact_index(True)
    yield
    ops.set_exact_index(was)


@pytest.mark.parametrize("shape", [(3, 5, 16), (2, 37, 9), (1, 64, 12), (2, 16, 1)])
def test_split_pair_layout_is_bit_exact(hific, dev, shape):
    """hific_split3 which = 2: source channel 16 g + j -> channels 32 g + j (hi) and 32 g + 16 + j (lo) of [outer][2 * C16]
    [inner]; padding channels of the last group zero.  Also what hific_channelnorm_fwd_exact(split_layout = 2) emits."""
    from hific_amd import lib, ops
    outer, C, inner = shape
    src = (_rnd(shape, 5) * 3.0).to(dev)
    hi = src.bfloat16().float()
    lo = (src - hi).bfloat16().float()
    C16 = (C + 15) // 16 * 16
    want = torch.zeros((outer, 2 * C16, inner), device=dev)
    for c in range(C):
        want[:, 32 * (c // 16) + c % 16] = hi[:, c]
        want[:, 32 * (c // 16) + 16 + c % 16] = lo[:, c]
    assert ops.pair_channels(C) == 2 * C16
    for dt, code in ((torch.bfloat16, lib.HIFIC_BF16), (torch.float32, lib.HIFIC_F32)):
        dst = torch.full((outer, 2 * C16, inner), 7.0, dtype=dt, device=dev)
        lib.call("hific_split3", src.data_ptr(), dst.data_ptr(), outer, C, inner, 2, code, lib.stream())
        torch.cuda.synchronize()
        assert torch.equal(dst.float(), want), dt
    if C >= 2:
        # the norm kernel's pair output = pair split of its float32 result (recomputed here from its own bf16 y + ... no:
        # from the 3C output of the same kernel, whose (hi, lo) halves are the same numbers)
        z = src.view(outer, C, inner, 1).contiguous()
        g = torch.rand(C, device=dev) + 0.5
        b = torch.rand(C, device=dev) - 0.5
        outs = {}
        for lay in (0, 2):
            Cx = 3 * C if lay == 0 else 2 * C16
            zb = torch.empty((outer, C, inner), dtype=torch.bfloat16, device=dev)
            y = torch.empty_like(zb)
            x3 = torch.full((outer, Cx, inner), 7.0, dtype=torch.bfloat16, device=dev)
            mean = torch.empty((outer, inner), device=dev); rstd = torch.empty_like(mean)
            lib.call("hific_channelnorm_fwd_exact", z.data_ptr(), g.data_ptr(), b.data_ptr(), zb.data_ptr(), y.data_ptr(),
                     x3.data_ptr(), mean.data_ptr(), rstd.data_ptr(), outer, C, inner, 1e-3, 1, lay, lib.stream())
            torch.cuda.synchronize()
            outs[lay] = (y.clone(), x3.float())
        assert torch.equal(outs[0][0], outs[2][0])
        h3, l3 = outs[0][1][:, :C], outs[0][1][:, C:2 * C]
        want2 = torch.zeros((outer, 2 * C16, inner), device=dev)
        for c in range(C):
            want2[:, 32 * (c // 16) + c % 16] = h3[:, c]
            want2[:, 32 * (c // 16) + 16 + c % 16] = l3[:, c]
        assert torch.equal(outs[2][1], want2)


@pytest.mark.parametrize("shape", [(3, 5, 16), (2, 7, 9), (1, 4, 4 * 33), (5, 1, 1)])
def test_split3_kernel_is_bit_exact(hific, dev, shape):
    """hi = bf16(v), lo = bf16(v - hi); activation layout (hi, lo, hi), weight layout (hi, hi, lo); bf16 and f32 outputs,
    vector (inner % 4 == 0) and scalar paths."""
    from hific_amd import lib
    outer, C, inner = shape
    src = (_rnd(shape, 1) * 3.0).to(dev)
    src.view(-1)[0] = 0.0
    hi = src.bfloat16()
    lo = (src - hi.float()).bfloat16()
    for which, parts in ((0, (hi, lo, hi)), (1, (hi, hi, lo))):
        want = torch.cat(parts, dim=1)
        for dt, code in ((torch.bfloat16, lib.HIFIC_BF16), (torch.float32, lib.HIFIC_F32)):
            dst = torch.full((outer, 3 * C, inner), 7.0, dtype=dt, device=dev)
            lib.call("hific_split3", src.data_ptr(), dst.data_ptr(), outer, C, inner, which, code, lib.stream())
            torch.cuda.synchronize()
            assert torch.equal(dst.float(), want.float()), (which, dt)
    # the two halves carry 16 significant bits: |v - (hi + lo)| <= 2^-17 |v|
    assert float(((hi.float() + lo.float()) - src).abs().max()) <= 2.0 ** -17 * float(src.abs().max())


@pytest.mark.parametrize("layout", [True, "pair"])
@pytest.mark.parametrize("name", list(CONV_CASES))
def test_exact_conv2d(hific, dev, name, layout):
    """layout True: (hi, lo, hi) x (hi, hi, lo) over 3C channels of the plain kernels; "pair": both operands in the pair
    layout (hific_split3 which = 2), cross terms formed by the native split kernel (gconv_kernel SPLIT)."""
    from hific_amd import ops, lib
    N, C, H, W, K, R, stride, pads, mode = CONV_CASES[name]
    pt, pl, pb, pr = pads
    x = _rnd((N, C, H, W), 1)
    w = _rnd((K, C, R, R), 2) * (1.0 / (C * R * R) ** 0.5)
    b = _rnd((K,), 3) * 0.1
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    xp = F.pad(xr, (pl, pr, pt, pb), mode="reflect" if mode == "reflect" else "constant")
    yr = F.relu(F.conv2d(xp, wr, br, stride=stride))
    gy = _rnd(tuple(yr.shape), 4)
    yr.backward(gy.double())
    xd, wd, bd = (t.to(dev).requires_grad_(True) for t in (x, w, b))
    pm = lib.PAD_REFLECT if mode == "reflect" else lib.PAD_ZERO
    y = ops.conv2d(xd, wd, bd, stride=stride, pads=pads, pad_mode=pm, act="relu", exact=layout)
    assert y.dtype == torch.float32 and y.shape == yr.shape
    y.backward(gy.to(dev))
    torch.cuda.synchronize()
    e_y = _relerr(y.detach().cpu(), yr.detach())
    # the same layer through the plain bf16 forward, for the printed comparison
    y16 = ops.conv2d(xd.detach(), wd.detach(), bd.detach(), stride=stride, pads=pads, pad_mode=pm, act="relu", out_f32=True)
    e_16 = _relerr(y16.cpu(), yr.detach())
    e_dx, e_dw, e_db = (_relerr(a.grad.cpu(), r.grad) for a, r in ((xd, xr), (wd, wr), (bd, br)))
    print(f"{name} [{layout}]: exact fwd {e_y:.2e} (plain bf16 {e_16:.2e}); bwd dx {e_dx:.2e} dw {e_dw:.2e} db {e_db:.2e}")
    assert e_y < FWD_TOL, e_y
    assert e_dx < BWD_TOL and e_dw < BWD_TOL and e_db < BWD_TOL, (e_dx, e_dw, e_db)


@pytest.mark.parametrize("layout", [True, "pair"])
@pytest.mark.parametrize("name", list(CONVT_CASES))
def test_exact_conv_transpose2d(hific, dev, name, layout):
    from hific_amd import ops
    N, Ci, H, W, Co, R, stride, pad, outpad = CONVT_CASES[name]
    x = _rnd((N, Ci, H, W), 1)
    w = _rnd((Ci, Co, R, R), 2) * (1.0 / (Ci * R * R) ** 0.5)
    b = _rnd((Co,), 3) * 0.1
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    yr = F.conv_transpose2d(xr, wr, br, stride=stride, padding=pad, output_padding=outpad)
    gy = _rnd(tuple(yr.shape), 4)
    yr.backward(gy.double())
    xd, wd, bd = (t.to(dev).requires_grad_(True) for t in (x, w, b))
    y = ops.conv_transpose2d(xd, wd, bd, stride, pad, outpad, exact=layout)
    assert y.dtype == torch.float32 and y.shape == yr.shape
    y.backward(gy.to(dev))
    torch.cuda.synchronize()
    e_y = _relerr(y.detach().cpu(), yr.detach())
    e_dx, e_dw, e_db = (_relerr(a.grad.cpu(), r.grad) for a, r in ((xd, xr), (wd, wr), (bd, br)))
    print(f"{name}: exact fwd {e_y:.2e}; bwd dx {e_dx:.2e} dw {e_dw:.2e} db {e_db:.2e}")
    assert e_y < FWD_TOL, e_y
    assert e_dx < BWD_TOL and e_dw < BWD_TOL and e_db < BWD_TOL, (e_dx, e_dw, e_db)


def test_split_weight_images_follow_the_optimizer(hific, dev):
    """The derived (hi, hi, lo) images and their packed copies are refreshed after a torch in-place update (version
    counter), after a FusedAdam step (arena epoch) and after the parameter was moved away and re-bound."""
    from hific_amd import ops, lib, optim
    ops.pack_cache.clear(); ops.split_weights.clear()
    torch.manual_seed(0)
    w1 = torch.nn.Parameter((torch.rand(96, 64, 3, 3) * 2 - 1).div(24).to(dev))
    w2 = torch.nn.Parameter((torch.rand(96, 48, 5, 5) * 2 - 1).div(40).to(dev))
    opt = optim.FusedAdam([w1, w2], lr=1e-2)
    x = (torch.rand(2, 64, 12, 12, device=dev) * 2 - 1).requires_grad_(True)

    def run():
        x.grad = None
        h = ops.conv2d(x, w1, None, stride=1, pads=(1, 1, 1, 1), pad_mode=lib.PAD_REFLECT, exact=True)
        y = ops.conv_transpose2d(h, w2, None, 2, 2, 1, exact=True)
        y.square().sum().backward()
        torch.cuda.synchronize()
        return y.detach().clone()

    def want():
        with torch.no_grad():
            h = F.conv2d(F.pad(x.detach().double().cpu(), (1, 1, 1, 1), mode="reflect"), w1.detach().double().cpu())
            return F.conv_transpose2d(h, w2.detach().double().cpu(), stride=2, padding=2, output_padding=1)

    assert _relerr(run().cpu(), want()) < 1e-4
    assert _relerr(run().cpu(), want()) < 1e-4            # cached
    with torch.no_grad():
        w1.mul_(1.5)
    opt.zero_grad()
    assert _relerr(run().cpu(), want()) < 1e-4            # version counter
    opt.step(); opt.zero_grad()
    assert _relerr(run().cpu(), want()) < 1e-4            # arena epoch
    ops.pack_cache.clear(); ops.split_weights.clear()


def _load(module, sd, prefix):
    module.load_state_dict({k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}, strict=True)
    return module


def test_encoder_exact_matches_oracle_to_1e4(hific, dev):
    from hific_amd.network.encoder import Encoder
    sd = O.make_state_dict(seed=0, gan=False, n_res=2)
    enc = _load(Encoder((3, 128, 128), 2, C=220), sd, "Encoder.").to(dev)
    x = O.make_image(1, 2, 128, 128)
    with torch.no_grad():
        yr = O.encoder_forward(sd, x)
        y = enc(x.to(dev))
    assert y.dtype == torch.float32
    e = _relerr(y.cpu(), yr)
    print(f"Encoder, bf16 + exact-index chain vs oracle f32: {e:.2e}")
    assert e < 1e-4


@pytest.mark.parametrize("training", [True, False])
def test_hyperprior_bf16_indices_equal_the_oracle(hific, dev, training):
    """bf16 compute mode: the quantised latent indices equal the oracle's except within float32 summation noise of a
    rounding tie (the assertion of test_gpu_modules.py::test_hyperprior_fp32), the six rates within 1e-3.  In eval mode the
    means come from ROUNDED hyperlatents (src/hyperprior.py:297-300), so an image is only compared if none of the oracle's
    own hyperlatents sits within 1e-4 of a rounding tie (seed 8: none does)."""
    from hific_amd.hyperprior import Hyperprior
    sd = O.make_state_dict(seed=0, gan=False, n_res=2)
    hp = _load(Hyperprior(bottleneck_capacity=220), sd, "Hyperprior.").to(dev).train(training)
    B, S = (4, 16) if training else (2, 8)
    y = O.make_noise(5 if training else 8, (B, 220, S, S)) * 6
    nh, nl = O.make_noise(6, (B, 320, S // 4, S // 4)), O.make_noise(7, (B, 220, S, S))
    with torch.no_grad():
        hr = O.hyperprior_forward(sd, y, (S * 16, S * 16), training, nh, nl)
        noises = [nh.to(dev), nl.to(dev)]
        hp._draw_noise = lambda t: noises.pop(0)
        h = hp(y.to(dev), (S * 16, S * 16))
    torch.cuda.synchronize()
    if not training:
        z = hr.hyperlatents + 0.5
        z = z - torch.floor(z)
        assert float(torch.minimum(z, 1 - z).min()) > 1e-4, "pick another seed: an oracle hyperlatent sits at a rounding tie"
    for f in ("latent_nbpp", "hyperlatent_nbpp", "total_nbpp", "latent_qbpp", "hyperlatent_qbpp", "total_qbpp"):
        a, b = float(getattr(h, f)), float(getattr(hr, f))
        assert abs(a - b) < 1e-3 * abs(b), (f, a, b)
    idx_o = O.quantized_indices(y, hr.latent_means)
    idx_h = torch.round(h.decoded.cpu() - hr.latent_means).to(torch.int64)
    flips = idx_h != idx_o
    n = int(flips.sum())
    print(f"hyperprior bf16 exact-index (training={training}): {n} of {flips.numel()} indices differ")
    if n:
        frac = y - hr.latent_means + 0.5
        frac = frac - torch.floor(frac)
        tie = torch.minimum(frac, 1 - frac)
        assert n <= max(2, 1e-4 * flips.numel()) and float(tie[flips].max()) < 1e-4, (n, float(tie[flips].max()))
        assert int(((idx_h - idx_o).abs() > 1).sum()) == 0


def test_encoder_falls_back_to_plain_bf16_beyond_the_split_image_limit(hific, dev, monkeypatch):
    """hific_channelnorm_fwd_exact / hific_split3 return HIFIC_ERR_UNSUPPORTED from 3*C*H*W >= 2^31 on
    (~11.9 MP for the 60-channel block).  The Encoder must then run the plain bf16 chain (with a warning) instead of
    raising; checked with the limit lowered so a 64x64 image trips it: the result equals the HIFIC_EXACT_INDEX=0 forward
    bit for bit, and the real limit itself is what the C-ABI reports."""
    import warnings
    from hific_amd import ops, lib
    from hific_amd.network import encoder
    torch.manual_seed(3)
    enc = encoder.Encoder((3, 64, 64), 2, C=220).to(dev)
    x = torch.rand(2, 3, 64, 64, device=dev)
    with torch.no_grad():
        y_exact = enc(x)
        ops.set_exact_index(False)
        y_plain = enc(x)
        ops.set_exact_index(True)
        monkeypatch.setattr(ops, "_EXACT_MAX_ELEMS", 3 * 60 * 64 * 64)          # block 1's output plane no longer fits
        monkeypatch.setattr(encoder, "_warned_big", False)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            y_fall = enc(x)
        assert any("plain bf16 chain" in str(i.message) for i in w)
        assert ops.exact_index_on()                                             # the suspension is scoped to the call
    assert torch.equal(y_fall, y_plain) and not torch.equal(y_fall, y_exact)
    # the limit the predicate mirrors: the C-ABI refuses 3*C*HW >= 2^31 without touching memory
    rc = lib.raw("hific_channelnorm_fwd_exact")(x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(),
                                                x.data_ptr(), x.data_ptr(), x.data_ptr(), 1, 60, (1 << 31) // 180 + 1,
                                                1e-3, 1, 0, lib.stream())
    assert rc == -4
    monkeypatch.undo()
    assert not ops.exact_chain_fits([(60, 3456, 3456)]) and ops.exact_chain_fits([(60, 3000, 3000)])

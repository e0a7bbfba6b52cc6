"""GPU parity of the convolution engine (csrc/gconv.hip) through the C-ABI: forward, data-gradient and
weight-gradient of every conv / conv-transpose geometry on the HiFIC hot path, against torch CPU float32 ops
(the arithmetic the reference executes: nn.Conv2d / nn.ConvTranspose2d / ReflectionPad2d).
Tolerances: float32 mode 2e-4 of the output scale (f32 MFMA is an exact fma chain; only the summation order
differs from oneDNN); bf16 mode 2e-2 (inputs are pre-rounded to bf16 on both sides, f32 accumulate)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# name: (N, C, H, W, K, R, stride, (pt, pl, pb, pr), mode)
CONV_CASES = {
    "E1_7x7_reflect3":      (2, 3, 64, 64, 60, 7, 1, (3, 3, 3, 3), "reflect"),
    "E2_3x3s2_asym":        (2, 60, 64, 64, 120, 3, 2, (1, 0, 0, 1), "reflect"),
    "E5_3x3s2_asym_480":    (2, 480, 32, 32, 960, 3, 2, (1, 0, 0, 1), "reflect"),
    "R_3x3_960":            (2, 960, 16, 16, 960, 3, 1, (1, 1, 1, 1), "reflect"),
    "E6_960_220":           (2, 960, 16, 16, 220, 3, 1, (1, 1, 1, 1), "reflect"),
    "A1_zero":              (2, 220, 16, 16, 320, 3, 1, (1, 1, 1, 1), "zeros"),
    "A2_5x5s2_reflect2":    (2, 320, 16, 16, 320, 5, 2, (2, 2, 2, 2), "reflect"),
    "A3_5x5s2_8x8":         (3, 320, 8, 8, 320, 5, 2, (2, 2, 2, 2), "reflect"),
    "G9_7x7_to3":           (2, 60, 64, 64, 3, 7, 1, (3, 3, 3, 3), "reflect"),
    "D1_4x4s2":             (2, 15, 64, 64, 64, 4, 2, (1, 1, 1, 1), "reflect"),
    "D4_4x4s2_256_512":     (2, 256, 32, 32, 512, 4, 2, (1, 1, 1, 1), "reflect"),
    "D5_1x1_to1":           (2, 512, 16, 16, 1, 1, 1, (0, 0, 0, 0), "zeros"),
    "L1_11x11s4":           (2, 3, 128, 128, 64, 11, 4, (2, 2, 2, 2), "zeros"),
    "L2_5x5":               (2, 64, 31, 31, 192, 5, 1, (2, 2, 2, 2), "zeros"),
    "L3_3x3_15":            (2, 192, 15, 15, 384, 3, 1, (1, 1, 1, 1), "zeros"),
    "odd_sizes":            (1, 5, 13, 17, 7, 3, 1, (1, 1, 1, 1), "reflect"),
    # gather-form reflect data gradient (gconv_sp9_kernel RFX): non-square, multi-tile, and shapes whose tiling forces
    # the padded-domain fall-back ((H-1) % TH == 0)
    "RFX_rect_12x20":       (3, 96, 12, 20, 64, 3, 1, (1, 1, 1, 1), "reflect"),
    "RFX_24x24_multi":      (1, 160, 24, 24, 128, 3, 1, (1, 1, 1, 1), "reflect"),
    "RFX_17x9":             (2, 64, 17, 9, 64, 3, 1, (1, 1, 1, 1), "reflect"),
    "RFX_4x4_min":          (5, 64, 4, 4, 96, 3, 1, (1, 1, 1, 1), "reflect"),
    "odd_s2":               (3, 9, 11, 10, 33, 3, 2, (1, 1, 1, 1), "zeros"),
    # the 128-row software-pipelined kernel (four reduction quarters) with zero padding, and on a narrow 12x8 plane
    "SP128_zero_16":        (2, 128, 16, 16, 256, 3, 1, (1, 1, 1, 1), "zeros"),
    "SP128_12x8":           (2, 64, 12, 8, 128, 3, 1, (1, 1, 1, 1), "reflect"),
    # weight gradient through the im2col kernel with 16 channel slots per tap (5..16 input channels; stride 1 and 2; D1 above
    # and odd_s2 take it too): wide plane (2 x 64 pixel tiles), narrow plane, zero and reflect padding
    "I16_3x3s1_c12":        (2, 12, 20, 28, 48, 3, 1, (1, 1, 1, 1), "reflect"),
    "I16_4x4s2_c16_zero":   (2, 16, 40, 24, 40, 4, 2, (1, 1, 1, 1), "zeros"),
    "I16_4x4s2_c15_wide":   (1, 15, 24, 256, 64, 4, 2, (1, 1, 1, 1), "reflect"),
    # phase-decomposed stride-2 weight gradient (wgrad_s2_kernel; E2 / E5 / D4 above take it too): partial pixel tiles
    # (OH = 6, 10), channel tails (70, 72, 40, 100), several column tiles, reflect and zero padding, both pad layouts
    "S2_3x3_rect_asym":     (3, 70, 12, 32, 72, 3, 2, (1, 0, 0, 1), "reflect"),
    "S2_3x3_zero_asym":     (2, 33, 16, 96, 65, 3, 2, (1, 0, 0, 1), "zeros"),
    "S2_3x3_sym_zero":      (2, 64, 16, 32, 64, 3, 2, (1, 1, 1, 1), "zeros"),
    "S2_4x4_reflect":       (2, 40, 20, 64, 100, 4, 2, (1, 1, 1, 1), "reflect"),
    "S2_4x4_zero":          (1, 130, 8, 32, 30, 4, 2, (1, 1, 1, 1), "zeros"),
    # natural-order stride-1 weight gradient (wgrad_s1_kernel; R_3x3_960 / E6 / A1 / SP128_zero_16 above take it too): several
    # column tiles, a partial tile row (H = 12, 20), channel tails, reflect and zero padding
    # pipelined stride-2 forward kernel (gconv_pl_kernel; E2 / E5 / D1 / D4 / S2_* above take it too): 32-pixel tile rows with a
    # partial last tile row (OH = 6, 10), several row tiles (K = 200), channel tails (C = 40, 70, 100), one workgroup walking
    # several tiles and channel chunks, reflect (rim columns on both sides) and zero padding, 3x3 and 4x4 windows
    "PL_3x3_wide_reflect":  (2, 70, 12, 128, 200, 3, 2, (1, 0, 0, 1), "reflect"),
    "PL_4x4_wide_reflect":  (3, 40, 20, 64, 100, 4, 2, (1, 1, 1, 1), "reflect"),
    "PL_3x3_sym_reflect":   (1, 100, 24, 64, 130, 3, 2, (1, 1, 1, 1), "reflect"),
    "PL_3x3_zero_many":     (9, 64, 32, 64, 64, 3, 2, (1, 1, 0, 0), "zeros"),
    # weight-resident persistent kernel of the few-channel layers on big planes (gconv_wr_kernel: <= 64 channels on both sides,
    # >= 1024 tiles of 256 pixels): the first Encoder layer's split form (9 channels, 7x7) with partial tiles in both
    # directions, the 60 -> 3 output layer (virtual-row form forward, virtual-channel form backward), the Discriminator's
    # first layer (stride 2, 15 channels), a plain 3x3 with zero padding and channel / row tails
    "WR_7x7_c9":            (4, 9, 250, 262, 60, 7, 1, (3, 3, 3, 3), "reflect"),
    "WR_7x7_c3":            (4, 3, 256, 248, 60, 7, 1, (3, 3, 3, 3), "reflect"),      # 3 channels: virtual-channel form (21 x 7 taps)
    "WR_7x7_to3":           (4, 60, 256, 256, 3, 7, 1, (3, 3, 3, 3), "reflect"),
    "WR_4x4s2_c15":         (5, 15, 520, 400, 64, 4, 2, (1, 1, 1, 1), "reflect"),
    "WR_3x3_c24_zero":      (3, 24, 300, 310, 50, 3, 1, (1, 1, 1, 1), "zeros"),
    # merged-phase kernel (gconv_mp_kernel) through the DATA GRADIENTS of stride-2 convolutions on big planes (the small cases
    # above take the per-phase grid): 4x4 reflect pad 1 (element stores into the fold), 3x3 asymmetric reflect pad (pair stores
    # into the fold), zero padding (plain pair stores), channel tails and partial tiles
    "PM_4x4_reflect":       (12, 64, 128, 128, 128, 4, 2, (1, 1, 1, 1), "reflect"),
    "PM_3x3_asym_reflect":  (6, 60, 256, 256, 120, 3, 2, (1, 0, 0, 1), "reflect"),
    "PM_3x3_zero_tail":     (5, 40, 200, 272, 72, 3, 2, (1, 1, 1, 1), "zeros"),
    # natural-order weight gradient for <= 4 input channels on big planes (wgrad_c3_kernel): the first Encoder layer's shape, and
    # 4 channels / 5x5 / zero padding / a row count that does not divide the workgroups
    "WC3_7x7":              (4, 3, 256, 256, 60, 7, 1, (3, 3, 3, 3), "reflect"),
    "WC3_5x5_c4_zero":      (5, 4, 208, 240, 40, 5, 1, (2, 2, 2, 2), "zeros"),
    "S1_rect_reflect":      (2, 70, 12, 32, 100, 3, 1, (1, 1, 1, 1), "reflect"),
    "S1_wide_zero":         (1, 130, 20, 48, 40, 3, 1, (1, 1, 1, 1), "zeros"),
}
# name: (N, Ci, H, W, Co, R, stride, pad, outpad)
CONVT_CASES = {
    "U1_960_480":  (2, 960, 16, 16, 480, 3, 2, 1, 1),
    "U4_120_60":   (2, 120, 32, 32, 60, 3, 2, 1, 1),
    "S1_5x5s2":    (2, 320, 4, 4, 320, 5, 2, 2, 1),
    "S2_5x5s2_8":  (3, 320, 8, 8, 320, 5, 2, 2, 1),
    "S3_3x3s1":    (2, 320, 16, 16, 220, 3, 1, 1, 0),
    "odd":         (1, 7, 5, 6, 9, 3, 2, 1, 1),
    "S2T_rect":    (2, 70, 6, 16, 40, 3, 2, 1, 1),        # wgrad_s2_kernel, conv-transpose form (zero outside), partial tile rows
    "S2T_wide":    (1, 20, 9, 48, 130, 3, 2, 1, 1),
    # merged-phase kernel forward on big planes: the Generator's last up-convolution shape, and odd sizes with a channel tail
    "PM_U4":       (4, 120, 128, 128, 60, 3, 2, 1, 1),
    "PM_T_odd":    (5, 70, 100, 136, 40, 3, 2, 1, 1),
}
DTYPES = [torch.float32, torch.bfloat16]
TOL = {torch.float32: 2e-4, torch.bfloat16: 2e-2}


def _rnd(shape, seed, dt):
    g = torch.Generator().manual_seed(seed)
    t = torch.rand(shape, generator=g) * 2 - 1
    return t.to(dt).float() if dt == torch.bfloat16 else t


def _relerr(a, b):
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-20)


def _ref_conv(x, w, b, stride, pads, mode):
    pt, pl, pb, pr = pads
    xp = F.pad(x, (pl, pr, pt, pb), mode="reflect" if mode == "reflect" else "constant")
    return F.conv2d(xp, w, b, stride=stride)


@pytest.mark.parametrize("dt", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("name", list(CONV_CASES))
def test_conv2d_fwd_bwd(hific, dev, name, dt):
    from hific_amd import ops, lib
    N, C, H, W, K, R, stride, pads, mode = CONV_CASES[name]
    hific.set_compute_dtype(dt)
    x = _rnd((N, C, H, W), 1, dt)
    w = _rnd((K, C, R, R), 2, dt) * (1.0 / (C * R * R) ** 0.5)
    w = w.to(dt).float() if dt == torch.bfloat16 else w
    b = _rnd((K,), 3, torch.float32) * 0.1
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = _ref_conv(xr, wr, br, stride, pads, mode)
    gy = _rnd(tuple(yr.shape), 4, dt)
    yr.backward(gy)

    xd = x.to(dev).to(dt).requires_grad_(True)
    wd = w.to(dev).requires_grad_(True)
    bd = b.to(dev).requires_grad_(True)
    pm = lib.PAD_REFLECT if mode == "reflect" else lib.PAD_ZERO
    y = ops.conv2d(xd, wd, bd, stride=stride, pads=pads, pad_mode=pm)
    assert y.shape == yr.shape and y.dtype == dt
    y.backward(gy.to(dev).to(dt))
    torch.cuda.synchronize()
    tol = TOL[dt]
    e_y = _relerr(y.detach().float().cpu(), yr.detach())
    e_dx = _relerr(xd.grad.float().cpu(), xr.grad)
    e_dw = _relerr(wd.grad.cpu(), wr.grad)
    e_db = _relerr(bd.grad.cpu(), br.grad)
    print(f"{name} {dt}: y {e_y:.2e} dx {e_dx:.2e} dw {e_dw:.2e} db {e_db:.2e}")
    assert e_y < tol, f"fwd {e_y}"
    assert e_dx < tol, f"bwd_data {e_dx}"
    assert e_dw < tol, f"bwd_weight {e_dw}"
    assert e_db < tol, f"bias grad {e_db}"


@pytest.mark.parametrize("dt", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("name", list(CONVT_CASES))
def test_conv_transpose2d_fwd_bwd(hific, dev, name, dt):
    from hific_amd import ops
    N, Ci, H, W, Co, R, stride, pad, outpad = CONVT_CASES[name]
    hific.set_compute_dtype(dt)
    x = _rnd((N, Ci, H, W), 1, dt)
    w = _rnd((Ci, Co, R, R), 2, dt) * (1.0 / (Ci * R * R) ** 0.5)
    w = w.to(dt).float() if dt == torch.bfloat16 else w
    b = _rnd((Co,), 3, torch.float32) * 0.1
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.conv_transpose2d(xr, wr, br, stride=stride, padding=pad, output_padding=outpad)
    gy = _rnd(tuple(yr.shape), 4, dt)
    yr.backward(gy)
    xd = x.to(dev).to(dt).requires_grad_(True)
    wd = w.to(dev).requires_grad_(True)
    bd = b.to(dev).requires_grad_(True)
    y = ops.conv_transpose2d(xd, wd, bd, stride, pad, outpad)
    assert y.shape == yr.shape
    y.backward(gy.to(dev).to(dt))
    torch.cuda.synchronize()
    tol = TOL[dt]
    e_y = _relerr(y.detach().float().cpu(), yr.detach())
    e_dx = _relerr(xd.grad.float().cpu(), xr.grad)
    e_dw = _relerr(wd.grad.cpu(), wr.grad)
    e_db = _relerr(bd.grad.cpu(), br.grad)
    print(f"{name} {dt}: y {e_y:.2e} dx {e_dx:.2e} dw {e_dw:.2e} db {e_db:.2e}")
    assert e_y < tol and e_dx < tol and e_dw < tol and e_db < tol, (e_y, e_dx, e_dw, e_db)


@pytest.mark.parametrize("act", ["relu", "leaky_relu"])
def test_conv_fused_activation_and_mixed_io(hific, dev, act):
    """bf16 compute with float32 input and float32 output (the entropy-model boundary) + fused activation."""
    from hific_amd import ops, lib
    hific.set_compute_dtype(torch.bfloat16)
    x = _rnd((2, 24, 16, 16), 1, torch.bfloat16)
    w = _rnd((40, 24, 3, 3), 2, torch.bfloat16) * 0.1
    w = w.to(torch.bfloat16).float()
    b = _rnd((40,), 3, torch.float32) * 0.1
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    z = _ref_conv(xr, wr, b, 1, (1, 1, 1, 1), "zeros")
    yr = F.relu(z) if act == "relu" else F.leaky_relu(z, 0.2)
    gy = _rnd(tuple(yr.shape), 4, torch.float32)
    yr.backward(gy)
    xd = x.to(dev).requires_grad_(True)           # float32 activations into a bf16-compute conv
    wd = w.to(dev).requires_grad_(True)
    y = ops.conv2d(xd, wd, b.to(dev), 1, (1, 1, 1, 1), lib.PAD_ZERO, act=act, out_f32=True)
    assert y.dtype == torch.float32
    y.backward(gy.to(dev))
    torch.cuda.synchronize()
    assert _relerr(y.detach().cpu(), yr.detach()) < 2e-2
    assert _relerr(xd.grad.cpu(), xr.grad) < 3e-2
    assert _relerr(wd.grad.cpu(), wr.grad) < 3e-2


def test_weight_pack_cache_tracks_weight_updates(hific, dev):
    """Persistent packed-weight cache (ops.WeightPackCache): cached calls equal un-cached ones bit for bit - after the
    first (filling) call, after an in-place torch update (version counter) and after a FusedAdam step (arena epoch);
    stale entries of several layers/directions are re-packed by ONE batched launch."""
    from hific_amd import ops, lib, optim
    hific.set_compute_dtype(torch.bfloat16)
    ops.pack_cache.clear()
    torch.manual_seed(0)
    w1 = torch.nn.Parameter((torch.rand(96, 64, 3, 3) * 2 - 1).div(24).to(dev))      # 3x3 reflect (RFX data gradient)
    w2 = torch.nn.Parameter((torch.rand(96, 48, 3, 3) * 2 - 1).div(24).to(dev))      # conv-transpose 96 -> 48
    opt = optim.FusedAdam([w1, w2], lr=1e-2)
    x = (torch.rand(2, 64, 12, 12, device=dev) * 2 - 1).bfloat16().requires_grad_(True)

    def run():
        x.grad = None
        h = ops.conv2d(x, w1, None, stride=1, pads=(1, 1, 1, 1), pad_mode=lib.PAD_REFLECT)
        y = ops.conv_transpose2d(h, w2, None, 2, 1, 1)
        y.float().square().sum().backward()
        torch.cuda.synchronize()
        return y.detach().clone(), x.grad.detach().clone()

    def run_uncached():
        on = ops._PACK_CACHE_ON
        ops._PACK_CACHE_ON = False
        try:
            return run()
        finally:
            ops._PACK_CACHE_ON = on

    def same(a, b):
        return torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])

    ref = run_uncached()
    assert same(run(), ref)                     # filling call (state 1)
    assert len(ops.pack_cache.entries) == 4     # two layers x two directions
    assert same(run(), ref)                     # cached (state 2)
    with torch.no_grad():
        w1.mul_(1.5)                            # in-place torch update: version counter
    opt.zero_grad()
    ref2 = run_uncached()
    assert not torch.equal(ref2[0], ref[0])
    opt.zero_grad()
    assert same(run(), ref2)                    # one batched re-pack of the stale entries
    opt.zero_grad()
    run()                                       # gradients for the optimizer
    opt.step()                                  # raw-kernel update: arena epoch
    opt.zero_grad()
    ref3 = run_uncached()
    assert not torch.equal(ref3[0], ref2[0])
    opt.zero_grad()
    assert same(run(), ref3)
    ops.pack_cache.clear()


def test_merged_phase_kernel_equals_the_per_phase_kernel(hific, dev):
    """gconv_mp_kernel (round 4: the two column phases of an output-row parity in one workgroup, one staged patch, pair
    stores) against the per-phase generic kernel on a launch large enough to take it (conv-transpose 120 -> 60, 8 x 64 x 64:
    512 workgroups): same chunk / tap accumulation order per output element, so the results are bit-identical; also the
    data gradient of a stride-2 reflect-padded conv (fold epilogue through the padded float32 buffer)."""
    import os
    from hific_amd import ops, lib
    hific.set_compute_dtype(torch.bfloat16)
    x = _rnd((8, 120, 64, 64), 11, torch.bfloat16).to(dev).bfloat16()
    w = (_rnd((120, 60, 3, 3), 12, torch.float32) * 0.05).to(dev)
    b = _rnd((60,), 13, torch.float32).to(dev)
    dy = _rnd((8, 128, 64, 64), 14, torch.bfloat16).to(dev).bfloat16()
    w2 = (_rnd((128, 64, 4, 4), 15, torch.float32) * 0.05).to(dev)
    w3 = (_rnd((128, 64, 3, 3), 16, torch.float32) * 0.05).to(dev)
    outs = {}
    was = os.environ.get("HIFIC_MP")
    try:
        for mp in ("1", "0"):
            os.environ["HIFIC_MP"] = mp
            lib.call("hific_env_refresh")          # the planner caches its knobs
            ops.pack_cache.clear()
            with torch.no_grad():
                y = ops.conv_transpose2d(x, w, b, 2, 1, 1, act="relu")
            dx = torch.empty((8, 64, 128, 128), dtype=torch.bfloat16, device=dev)
            ws = lib.workspace(dev)
            lib.call("hific_conv2d_bwd_data", dy.data_ptr(), w2.data_ptr(), None, dx.data_ptr(), 8, 64, 128, 128, 128, 4, 4, 2,
                     1, 1, 1, 1, lib.PAD_REFLECT, lib.HIFIC_BF16, 0, ws.data_ptr(), ws.numel(), None, 0, 0, lib.stream())
            # the Encoder's asymmetric reflect pad (top 1, left 0, bottom 0, right 1), 3x3 stride 2: even left pad -> pair
            # stores straight into dx for the interior, the rim through the padded float32 buffer
            dx2 = torch.empty((8, 64, 128, 128), dtype=torch.bfloat16, device=dev)
            lib.call("hific_conv2d_bwd_data", dy.data_ptr(), w3.data_ptr(), None, dx2.data_ptr(), 8, 64, 128, 128, 128, 3, 3, 2,
                     1, 0, 0, 1, lib.PAD_REFLECT, lib.HIFIC_BF16, 0, ws.data_ptr(), ws.numel(), None, 0, 0, lib.stream())
            torch.cuda.synchronize()
            outs[mp] = (y.clone(), dx.clone(), dx2.clone())
    finally:
        if was is None:
            os.environ.pop("HIFIC_MP", None)
        else:
            os.environ["HIFIC_MP"] = was
        lib.call("hific_env_refresh")
        ops.pack_cache.clear()
    assert torch.equal(outs["1"][0], outs["0"][0])
    assert torch.equal(outs["1"][1], outs["0"][1])
    assert torch.equal(outs["1"][2], outs["0"][2])
    xr = torch.zeros((8, 64, 128, 128), dtype=torch.float64, requires_grad=True)
    F.conv2d(F.pad(xr, (0, 1, 1, 0), mode="reflect"), w3.double().cpu(), stride=2).backward(dy.double().cpu())
    assert _relerr(outs["1"][2].float().cpu(), xr.grad.float()) < 2e-2
    yr = F.relu(F.conv_transpose2d(x.float().cpu(), w.cpu(), b.cpu(), stride=2, padding=1, output_padding=1))
    assert _relerr(outs["1"][0].float().cpu(), yr) < 2e-2


def _kinds_of(fn):
    """Kernel kinds (in-library profiler names) launched by fn()."""
    import ctypes
    from hific_amd import lib
    lib.call("hific_prof_begin")
    fn()
    ms = (ctypes.c_double * 32)(); fl = (ctypes.c_double * 32)(); cnt = (ctypes.c_int * 32)()
    names = ctypes.create_string_buffer(32 * 64)
    nk = lib.raw("hific_prof_end")(32, ms, fl, cnt, names)
    assert nk >= 0
    return {names.raw[k * 64:(k + 1) * 64].split(bytes(1), 1)[0].decode() for k in range(nk) if cnt[k]}


def test_strided_weight_gradients_take_their_kernels(hific, dev):
    """The parity cases above would also pass on the generic kernel: pin the dispatch.  3x3 / 4x4 stride-2 layers with
    OW % 16 == 0 -> wgrad_s2_kernel (conv and conv-transpose form); 5..16 input channels -> the 16-slot im2col kernel."""
    from hific_amd import ops, lib
    hific.set_compute_dtype(torch.bfloat16)

    def conv_wgrad(name):
        N, C, H, W, K, R, stride, pads, mode = CONV_CASES[name]
        x = _rnd((N, C, H, W), 1, torch.bfloat16).to(dev).bfloat16()
        w = (_rnd((K, C, R, R), 2, torch.bfloat16) * 0.05).to(dev).requires_grad_(True)
        pm = lib.PAD_REFLECT if mode == "reflect" else lib.PAD_ZERO
        y = ops.conv2d(x, w, None, stride=stride, pads=pads, pad_mode=pm)
        gy = torch.ones_like(y)
        torch.cuda.synchronize()
        return _kinds_of(lambda: (y.backward(gy), torch.cuda.synchronize()))

    def convt_wgrad(name):
        N, Ci, H, W, Co, R, stride, pad, outpad = CONVT_CASES[name]
        x = _rnd((N, Ci, H, W), 1, torch.bfloat16).to(dev).bfloat16()
        w = (_rnd((Ci, Co, R, R), 2, torch.bfloat16) * 0.05).to(dev).requires_grad_(True)
        y = ops.conv_transpose2d(x, w, None, stride, pad, outpad)
        gy = torch.ones_like(y)
        torch.cuda.synchronize()
        return _kinds_of(lambda: (y.backward(gy), torch.cuda.synchronize()))

    for name in ("E2_3x3s2_asym", "E5_3x3s2_asym_480", "D4_4x4s2_256_512", "S2_3x3_rect_asym", "S2_4x4_zero", "D1_4x4s2",
                 "I16_4x4s2_c15_wide"):
        assert "wgrad_s2_kernel" in conv_wgrad(name), name
    for name in ("U1_960_480", "S2T_rect"):
        assert "wgrad_s2_kernel" in convt_wgrad(name), name
    for name in ("R_3x3_960", "E6_960_220", "A1_zero", "S1_rect_reflect", "S1_wide_zero"):
        assert "wgrad_s1_kernel" in conv_wgrad(name), name
    for name in ("I16_3x3s1_c12", "I16_4x4s2_c16_zero", "odd_s2"):
        assert "wgrad_im2col_kernel<bf16>" in conv_wgrad(name), name
    assert "wgrad_s2_kernel" not in conv_wgrad("odd_s2")
    for name in ("WC3_7x7", "WC3_5x5_c4_zero"):
        assert "wgrad_c3_kernel" in conv_wgrad(name), name
    assert "wgrad_c3_kernel" not in conv_wgrad("WR_7x7_c3")          # plane width not a multiple of 16: im2col kernel
    hific.set_compute_dtype(torch.float32)


def test_few_channel_layers_take_the_weight_resident_kernel(hific, dev):
    """Pins the dispatch of the few-channel big-plane layers to gconv_wr_kernel (forward and, where the transposed problem
    qualifies too, the data gradient), checks what it must decline (small planes, many channels), and that it agrees with the
    generic kernels (HIFIC_WR=0) to bf16 output rounding."""
    import os
    from hific_amd import ops, lib
    hific.set_compute_dtype(torch.bfloat16)

    def run(name, with_out=False, bwd=False):
        N, C, H, W, K, R, stride, pads, mode = CONV_CASES[name]
        x = _rnd((N, C, H, W), 1, torch.bfloat16).to(dev).bfloat16().requires_grad_(bwd)
        w = (_rnd((K, C, R, R), 2, torch.bfloat16) * 0.05).to(dev)
        b = _rnd((K,), 3, torch.float32).to(dev)
        pm = lib.PAD_REFLECT if mode == "reflect" else lib.PAD_ZERO
        out = {}
        def go():
            with torch.set_grad_enabled(bwd):
                y = ops.conv2d(x, w, b, stride=stride, pads=pads, pad_mode=pm, act=None if bwd else "leaky_relu")
            if bwd:
                y.backward(torch.ones_like(y))
                out["y"] = x.grad
            else:
                out["y"] = y
            torch.cuda.synchronize()
        kinds = _kinds_of(go)
        return (kinds, out["y"]) if with_out else kinds

    for name in ("WR_7x7_c9", "WR_7x7_c3", "WR_7x7_to3", "WR_4x4s2_c15", "WR_3x3_c24_zero"):
        kinds = run(name)
        assert any(k.startswith("gconv_wr_kernel") for k in kinds), (name, kinds)
    kinds = run("WR_3x3_c24_zero", bwd=True)                          # a data gradient that is a few-channel problem itself
    assert any(k.startswith("gconv_wr_kernel") for k in kinds), kinds
    for name in ("G9_7x7_to3", "D1_4x4s2", "E2_3x3s2_asym", "R_3x3_960"):
        kinds = run(name)
        assert not any(k.startswith("gconv_wr_kernel") for k in kinds), (name, kinds)
    was = os.environ.get("HIFIC_WR")
    try:
        for name, bwd in (("WR_7x7_c9", False), ("WR_7x7_c3", False), ("WR_7x7_to3", False), ("WR_4x4s2_c15", False),
                          ("WR_3x3_c24_zero", True)):
            _, y1 = run(name, with_out=True, bwd=bwd)
            os.environ["HIFIC_WR"] = "0"
            lib.call("hific_env_refresh"); ops.pack_cache.clear()
            kinds0, y0 = run(name, with_out=True, bwd=bwd)
            assert not any(k.startswith("gconv_wr_kernel") for k in kinds0)
            os.environ.pop("HIFIC_WR")
            lib.call("hific_env_refresh"); ops.pack_cache.clear()
            scale = y0.float().abs().max().item()
            assert (y1.float() - y0.float()).abs().max().item() <= 2.0 ** -7 * scale, (name, bwd)
    finally:
        if was is None:
            os.environ.pop("HIFIC_WR", None)
        else:
            os.environ["HIFIC_WR"] = was
        lib.call("hific_env_refresh"); ops.pack_cache.clear()
    hific.set_compute_dtype(torch.float32)


def test_stride2_forward_layers_take_the_pipelined_kernel(hific, dev):
    """Pins the dispatch of the stride-2 forward-type layers to gconv_pl_kernel (the parity cases would also pass on the
    generic kernel), checks what it must decline (odd planes, few output channels, float32 parity mode), and that it agrees with
    the generic kernel (HIFIC_PL=0) to bf16 output rounding - same products, different summation order inside a chunk."""
    import os
    from hific_amd import ops, lib
    hific.set_compute_dtype(torch.bfloat16)

    def fwd(name, with_out=False):
        N, C, H, W, K, R, stride, pads, mode = CONV_CASES[name]
        x = _rnd((N, C, H, W), 1, torch.bfloat16).to(dev).bfloat16()
        w = (_rnd((K, C, R, R), 2, torch.bfloat16) * 0.05).to(dev)
        b = _rnd((K,), 3, torch.float32).to(dev)
        pm = lib.PAD_REFLECT if mode == "reflect" else lib.PAD_ZERO
        out = {}
        def run():
            with torch.no_grad():
                out["y"] = ops.conv2d(x, w, b, stride=stride, pads=pads, pad_mode=pm, act="leaky_relu")
            torch.cuda.synchronize()
        kinds = _kinds_of(run)
        return (kinds, out["y"]) if with_out else kinds

    for name in ("E2_3x3s2_asym", "E5_3x3s2_asym_480", "D1_4x4s2", "D4_4x4s2_256_512", "S2_3x3_rect_asym", "S2_3x3_zero_asym",
                 "S2_4x4_reflect", "PL_3x3_wide_reflect", "PL_4x4_wide_reflect", "PL_3x3_sym_reflect", "PL_3x3_zero_many"):
        kinds = fwd(name)
        assert any(k.startswith("gconv_pl_kernel") for k in kinds), (name, kinds)
    for name in ("odd_s2", "S2_4x4_zero", "A2_5x5s2_reflect2", "R_3x3_960"):
        kinds = fwd(name)
        assert not any(k.startswith("gconv_pl_kernel") for k in kinds), (name, kinds)
    was = os.environ.get("HIFIC_PL")
    try:
        for name in ("PL_3x3_wide_reflect", "PL_4x4_wide_reflect", "D1_4x4s2"):
            _, y1 = fwd(name, with_out=True)
            os.environ["HIFIC_PL"] = "0"
            lib.call("hific_env_refresh"); ops.pack_cache.clear()
            kinds0, y0 = fwd(name, with_out=True)
            assert not any(k.startswith("gconv_pl_kernel") for k in kinds0)
            os.environ.pop("HIFIC_PL")
            lib.call("hific_env_refresh"); ops.pack_cache.clear()
            scale = y0.float().abs().max().item()
            assert (y1.float() - y0.float()).abs().max().item() <= 2.0 ** -7 * scale, name
    finally:
        if was is None:
            os.environ.pop("HIFIC_PL", None)
        else:
            os.environ["HIFIC_PL"] = was
        lib.call("hific_env_refresh"); ops.pack_cache.clear()
    hific.set_compute_dtype(torch.float32)

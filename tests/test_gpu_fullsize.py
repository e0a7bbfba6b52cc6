"""Size-independent properties at the BASELINE sizes (batch 16, 256x256, bf16 and f32), where the CPU oracle would
take too long: adjointness of every conv kernel triple (<y, A x> = <A^T y, x>, <dW, W'> = <y, conv(x; W')>),
linearity, and a full-size training step (finite losses / gradients, bf16 step within 2% of the f32 step,
deterministic replays)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dot(a, b):
    return float((a.double() * b.double()).sum())


CASES = {   # (N, C, H, W, K, R, stride, pads, mode): the Generator residual conv, a strided Encoder conv, the 7x7 head
    "resblock_960": (16, 960, 16, 16, 960, 3, 1, (1, 1, 1, 1), "reflect"),
    "encoder_s2_240_480": (16, 240, 64, 64, 480, 3, 2, (1, 0, 0, 1), "reflect"),
    "encoder_7x7_3_60": (16, 3, 256, 256, 60, 7, 1, (3, 3, 3, 3), "reflect"),
    "generator_7x7_60_3": (16, 60, 256, 256, 3, 7, 1, (3, 3, 3, 3), "reflect"),
    # the Discriminator's 4x4 stride-2 layers at the G+D batch (32 images), the hyper-analysis 5x5 stride 2, LPIPS 5x5
    "disc_4x4s2_15_64": (32, 15, 256, 256, 64, 4, 2, (1, 1, 1, 1), "reflect"),
    "disc_4x4s2_128_256": (32, 128, 64, 64, 256, 4, 2, (1, 1, 1, 1), "reflect"),
    "hyper_5x5s2_320": (16, 320, 16, 16, 320, 5, 2, (2, 2, 2, 2), "reflect"),
    "lpips_5x5_64_192": (32, 64, 31, 31, 192, 5, 1, (2, 2, 2, 2), "zeros"),
}
CASES_T = {   # (N, Ci, H, W, Co, R, stride, pad, outpad): Generator up-convolutions, hyper-synthesis 5x5 stride 2
    "up_960_480": (16, 960, 16, 16, 480, 3, 2, 1, 1),
    "up_120_60": (16, 120, 128, 128, 60, 3, 2, 1, 1),
    "synthesis_5x5s2_320": (16, 320, 8, 8, 320, 5, 2, 2, 1),
}


@pytest.mark.parametrize("dt,tol", [(torch.float32, 2e-4), (torch.bfloat16, 2e-2)], ids=["f32", "bf16"])
@pytest.mark.parametrize("name", list(CASES))
def test_conv_adjoint_identities_full_size(hific, dev, name, dt, tol):
    from hific_amd import ops, lib
    N, C, H, W, K, R, stride, pads, mode = CASES[name]
    hific.set_compute_dtype(dt)
    g = torch.Generator(device=dev).manual_seed(0)
    x = (torch.rand((N, C, H, W), generator=g, device=dev) * 2 - 1).to(dt)
    w = ((torch.rand((K, C, R, R), generator=g, device=dev) * 2 - 1) / (C * R * R) ** 0.5)
    w = w.to(dt).float()
    w2 = ((torch.rand((K, C, R, R), generator=g, device=dev) * 2 - 1) / (C * R * R) ** 0.5).to(dt).float()
    pm = lib.PAD_REFLECT if mode == "reflect" else lib.PAD_ZERO
    xg = x.clone().requires_grad_(True)
    wg = w.clone().requires_grad_(True)
    y = ops.conv2d(xg, wg, None, stride, pads, pm)
    gy = (torch.rand(y.shape, generator=g, device=dev) * 2 - 1).to(dt)
    y.backward(gy)
    # <gy, conv(x; w)> == <dx, x>   and   == <dw, w>  (the conv is bilinear in (x, w))
    lhs = _dot(gy, y.detach())
    assert abs(_dot(xg.grad, x) - lhs) <= tol * abs(lhs) + 1e-3, name
    assert abs(_dot(wg.grad, w) - lhs) <= tol * abs(lhs) + 1e-3, name
    # <dw, w2> == <gy, conv(x; w2)>
    y2 = ops.conv2d(x, w2, None, stride, pads, pm)
    rhs = _dot(gy, y2)
    assert abs(_dot(wg.grad, w2) - rhs) <= tol * max(abs(rhs), abs(lhs)) + 1e-3, name
    # linearity in x
    y3 = ops.conv2d((x.float() * 0.5).to(dt), w, None, stride, pads, pm)
    assert float((y3.float() - 0.5 * y.detach().float()).abs().max()) <= tol * float(y.detach().float().abs().max()) + 1e-6


@pytest.mark.parametrize("dt,tol", [(torch.float32, 2e-4), (torch.bfloat16, 2e-2)], ids=["f32", "bf16"])
@pytest.mark.parametrize("name", list(CASES_T))
def test_conv_transpose_adjoint_identities_full_size(hific, dev, name, dt, tol):
    """Same identities for the transposed convolutions at the benchmarked sizes (sub-pixel phase kernels, merged-phase
    sp9 kernel, strided weight-gradient plans): <gy, convT(x; w)> = <dx, x> = <dw, w>, <dw, w2> = <gy, convT(x; w2)>."""
    from hific_amd import ops
    N, Ci, H, W, Co, R, stride, pad, outpad = CASES_T[name]
    hific.set_compute_dtype(dt)
    g = torch.Generator(device=dev).manual_seed(0)
    x = (torch.rand((N, Ci, H, W), generator=g, device=dev) * 2 - 1).to(dt)
    w = ((torch.rand((Ci, Co, R, R), generator=g, device=dev) * 2 - 1) / (Ci * R * R) ** 0.5).to(dt).float()
    w2 = ((torch.rand((Ci, Co, R, R), generator=g, device=dev) * 2 - 1) / (Ci * R * R) ** 0.5).to(dt).float()
    xg, wg = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y = ops.conv_transpose2d(xg, wg, None, stride, pad, outpad)
    gy = (torch.rand(y.shape, generator=g, device=dev) * 2 - 1).to(dt)
    y.backward(gy)
    lhs = _dot(gy, y.detach())
    assert abs(_dot(xg.grad, x) - lhs) <= tol * abs(lhs) + 1e-3, name
    assert abs(_dot(wg.grad, w) - lhs) <= tol * abs(lhs) + 1e-3, name
    y2 = ops.conv_transpose2d(x, w2, None, stride, pad, outpad)
    rhs = _dot(gy, y2)
    assert abs(_dot(wg.grad, w2) - rhs) <= tol * max(abs(rhs), abs(lhs)) + 1e-3, name
    y3 = ops.conv_transpose2d((x.float() * 0.5).to(dt), w, None, stride, pad, outpad)
    assert float((y3.float() - 0.5 * y.detach().float()).abs().max()) <= tol * float(y.detach().float().abs().max()) + 1e-6


def _two_steps_overlap(hific, dev, overlap):
    """Two training steps (the second one uses weights and packs produced by the first optimizer step)."""
    import hific_amd
    from hific_amd import optim
    from hific_amd.default_config import make_args, mse_lpips_args, ModelTypes
    hific.set_compute_dtype(torch.bfloat16)
    torch.manual_seed(5)
    model = hific_amd.Model(make_args(mse_lpips_args, batch_size=8), model_type=ModelTypes.COMPRESSION,
                            allow_random_lpips_backbone=True, device_rate_select=True).to(dev).train()
    amort_params = [p for m in model.amortization_models for p in m.parameters()]
    n_enc = len(list(model.Encoder.parameters()))
    amort = optim.FusedAdam(amort_params, lr=1e-3, overlap_from=n_enc if overlap else None)
    hyper = optim.FusedAdam(list(model.Hyperprior.hyperlatent_likelihood.parameters()), lr=1e-3)
    g = torch.Generator(device=dev).manual_seed(6)
    out = []
    for it in range(2):
        x = torch.rand((8, 3, 256, 256), generator=g, device=dev)
        torch.manual_seed(7 + it)
        losses = model(x, train_generator=True, writeout=False)
        losses["compression"].backward()
        out.append(float(losses["compression"]))
        amort.step(); hyper.step(); amort.zero_grad(); hyper.zero_grad()
    amort.synchronize()
    torch.cuda.synchronize()
    return out, amort.arena.flat.clone()


def test_optimizer_tail_on_its_own_stream_is_bit_identical(hific, dev):
    """FusedAdam(overlap_from=#Encoder parameters): Generator / hyper-net parameters and their re-packs are updated on the
    optimizer stream under the next Encoder forward (model.py waits before it touches them).  Losses of both steps and
    the final parameters must equal the single-stream run bit for bit."""
    from hific_amd import ops
    was = ops._OPT_STREAM_ON
    ops._OPT_STREAM_ON = True               # opt-in feature (HIFIC_OPT_STREAM=1)
    try:
        l0, p0 = _two_steps_overlap(hific, dev, False)
        l1, p1 = _two_steps_overlap(hific, dev, True)
    finally:
        ops._OPT_STREAM_ON = was
    assert l0 == l1, (l0, l1)
    assert torch.equal(p0, p1)


def _one_step(hific, dev, dt, seed):
    import hific_amd
    from hific_amd import optim
    from hific_amd.default_config import make_args, mse_lpips_args, ModelTypes
    hific.set_compute_dtype(dt)
    torch.manual_seed(seed)
    model = hific_amd.Model(make_args(mse_lpips_args, batch_size=16), model_type=ModelTypes.COMPRESSION,
                            allow_random_lpips_backbone=True,
                            device_rate_select=True).to(dev).train()
    amort = optim.FusedAdam([p for m in model.amortization_models for p in m.parameters()], lr=1e-4)
    hyper = optim.FusedAdam(list(model.Hyperprior.hyperlatent_likelihood.parameters()), lr=1e-4)
    g = torch.Generator(device=dev).manual_seed(seed + 1)
    x = torch.rand((16, 3, 256, 256), generator=g, device=dev)
    torch.manual_seed(seed + 2)          # noise draws
    losses = model(x, train_generator=True, writeout=False)
    losses["compression"].backward()
    loss = float(losses["compression"])
    gflat = amort.arena.flat_grad.clone()
    hflat = hyper.arena.flat_grad.clone()
    amort.step(); hyper.step()
    torch.cuda.synchronize()
    return loss, gflat, hflat, amort.arena.flat.clone()


def test_full_size_training_step_properties(hific, dev):
    l32, g32, h32, p32 = _one_step(hific, dev, torch.float32, 0)
    l16, g16, h16, p16 = _one_step(hific, dev, torch.bfloat16, 0)
    l16b, g16b, _, p16b = _one_step(hific, dev, torch.bfloat16, 0)
    for t in (g32, h32, g16, h16, p32, p16):
        assert torch.isfinite(t).all()
    assert abs(l16 - l32) <= 0.02 * abs(l32), (l16, l32)
    # run-to-run determinism (fixed summation orders everywhere, no atomics)
    assert l16 == l16b and torch.equal(g16, g16b) and torch.equal(p16, p16b)
    # gradient direction of the bf16 step agrees with the f32 step
    cos = float((g16.double() * g32.double()).sum() / (g16.double().norm() * g32.double().norm()))
    assert cos > 0.98, cos
    # Adam moved every parameter that has a gradient by at most lr
    assert float((p16 - p16b).abs().max()) == 0.0


def test_side_stream_weight_gradients_equal_single_stream(hific, dev):
    """ops._SideLaunch: weight / bias gradients of arena parameters run on a second stream.  Same kernels, same
    summation orders: the step must be bit-identical to single-stream execution, and the gradients must be complete
    when backward() returns (they are read right after it, on the main stream, without a device synchronisation)."""
    from hific_amd import ops
    was = ops._SIDE_ON
    try:
        ops.set_side_stream(False)
        l0, g0, h0, p0 = _one_step(hific, dev, torch.bfloat16, 3)
        ops.set_side_stream(True)
        l1, g1, h1, p1 = _one_step(hific, dev, torch.bfloat16, 3)
    finally:
        ops.set_side_stream(was)
    assert l0 == l1 and torch.equal(g0, g1) and torch.equal(h0, h1) and torch.equal(p0, p1)

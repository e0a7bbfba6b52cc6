"""The Generator as an exact chain (round 6; ops.set_exact_training / ops.set_exact_reconstruction in bf16 mode):
network/generator.py::_forward_exact_chain runs every conv -> ChannelNorm block as one op whose contraction reads split-bf16
operands, whose norm kernel emits the nominal bf16 activation + the next block's split image, and whose residual / head-skip
sums are formed from the split images.  The autograd graph is the plain bf16 one.  Checked here against the oracle's
Generator (src/network/generator.py:9-44,98-168):
  * the two new kernels bit for bit / to float32 rounding (hific_channelnorm_fwd_exact_res, hific_add_split),
  * the module forward within north_star's 1e-3 (measured ~2e-5) where plain bf16 gives ~1e-2,
  * every gradient of the module at the bf16 tolerance of test_gpu_modules.py::test_generator, and equal (within bf16
    operand distance) to the plain bf16 mode's,
  * the no-grad (exact-reconstruction) route gives the same values as the training route."""
import pytest
import torch

from oracle import hific_oracle as O
from gradcheck import check_grads

pytestmark = pytest.mark.gpu

N_RES = 2


def _relerr(a, b):
    return (a.double() - b.double()).abs().max().item() / max(b.abs().max().item(), 1e-20)


def _bf(x):
    return x.to(torch.bfloat16).float()


def _split_cpu(v, layout):
    """float32 [N,C,HW] -> bf16 split image as hific_split3 lays it out (0: (hi, lo, hi); 2: pair groups)."""
    hi = v.to(torch.bfloat16)
    lo = (v - hi.float()).to(torch.bfloat16)
    N, C, HW = v.shape
    if layout == 0:
        return torch.cat([hi, lo, hi], dim=1)
    C16 = (C + 15) // 16 * 16
    out = torch.zeros(N, 2 * C16, HW, dtype=torch.bfloat16)
    for c in range(C):
        out[:, 32 * (c // 16) + (c % 16)] = hi[:, c]
        out[:, 32 * (c // 16) + 16 + (c % 16)] = lo[:, c]
    return out


@pytest.fixture(autouse=True)
def _bf16_mode(hific):
    from hific_amd import ops
    hific.set_compute_dtype(torch.bfloat16)
    yield
    ops.set_exact_training(False)
    ops.set_exact_reconstruction(False)
    ops.set_exact_generator_fused(True)


@pytest.mark.parametrize("N,C,HW,relu,lay_out,lay_res", [(2, 960, 256, 0, 0, 0), (1, 37, 100, 1, 2, 0), (2, 60, 4096, 0, 0, 2),
                                                         (3, 220, 64, 1, 2, 2)])
def test_channelnorm_exact_res_kernel(hific, dev, N, C, HW, relu, lay_out, lay_res):
    """y = relu?(norm(z)) + (r_hi + r_lo): float32 arithmetic on the device vs the same formula in float64 on the host;
    zb = bf16(z) bit for bit; y3 = the split image of the device's own float32 result (hi bit for bit with y)."""
    from hific_amd import lib
    g = torch.Generator().manual_seed(5)
    z = torch.randn(N, C, HW, generator=g) * 3 + 0.5
    r = torch.randn(N, C, HW, generator=g) * 2
    gamma = torch.rand(C, generator=g) + 0.5
    beta = torch.randn(C, generator=g) * 0.3
    r3 = _split_cpu(r, lay_res)
    r_eff = _bf(r) + (r - _bf(r)).to(torch.bfloat16).float()           # what hi + lo carries
    Cx = 3 * C if lay_out == 0 else 2 * ((C + 15) // 16 * 16)
    zd, r3d, gd, bd = z.to(dev), r3.to(dev), gamma.to(dev), beta.to(dev)
    zb = torch.empty(N, C, HW, dtype=torch.bfloat16, device=dev)
    y = torch.empty_like(zb)
    y3 = torch.full((N, Cx, HW), 7.0, dtype=torch.bfloat16, device=dev)
    mean = torch.empty(N, HW, device=dev)
    rstd = torch.empty(N, HW, device=dev)
    lib.call("hific_channelnorm_fwd_exact_res", zd.data_ptr(), gd.data_ptr(), bd.data_ptr(), r3d.data_ptr(), lay_res,
             zb.data_ptr(), y.data_ptr(), y3.data_ptr(), mean.data_ptr(), rstd.data_ptr(), N, C, HW, 1e-3, relu, lay_out,
             lib.stream())
    torch.cuda.synchronize()
    z64 = z.double()
    mu = z64.mean(1, keepdim=True)
    var = ((z64 - mu) ** 2).sum(1, keepdim=True) / (C - 1)
    o = gamma.double()[None, :, None] * (z64 - mu) * torch.rsqrt(var + 1e-3) + beta.double()[None, :, None]
    if relu:
        o = o.clamp_min(0)
    ref = o + r_eff.double()
    assert torch.equal(zb.cpu(), z.to(torch.bfloat16))
    y_c, y3_c = y.float().cpu(), y3.float().cpu()
    assert _relerr(y_c, ref.float()) < 5e-3                               # bf16 rounding of the nominal output
    if lay_out == 0:
        hi, lo, hi2 = y3_c[:, :C], y3_c[:, C:2 * C], y3_c[:, 2 * C:]
        assert torch.equal(hi, hi2)
    else:
        idx_h = torch.tensor([32 * (c // 16) + (c % 16) for c in range(C)])
        hi, lo = y3_c[:, idx_h], y3_c[:, idx_h + 16]
        pad = torch.ones(Cx, dtype=torch.bool)
        pad[idx_h] = False; pad[idx_h + 16] = False
        assert float(y3_c[:, pad].abs().max()) == 0.0 if pad.any() else True
    assert torch.equal(hi, y_c)
    assert (hi.double() + lo.double() - ref).abs().max().item() < 3e-5 * ref.abs().max().item()
    assert (mean.cpu().double() - mu[:, 0]).abs().max().item() < 1e-5 * (1 + mu.abs().max().item())


@pytest.mark.parametrize("N,C,HW,la,lb,lo", [(2, 960, 256, 0, 0, 2), (1, 37, 100, 2, 0, 0), (2, 16, 33, 2, 2, 2)])
def test_add_split_kernel(hific, dev, N, C, HW, la, lb, lo):
    from hific_amd import lib
    g = torch.Generator().manual_seed(6)
    a, b = torch.randn(N, C, HW, generator=g), torch.randn(N, C, HW, generator=g) * 5
    a3, b3 = _split_cpu(a, la), _split_cpu(b, lb)

    def eff(v):
        return _bf(v) + (v - _bf(v)).to(torch.bfloat16).float()
    ref = (eff(a) + eff(b))                                              # float32 sum of two 16-bit-mantissa values
    Cy = 3 * C if lo == 0 else 2 * ((C + 15) // 16 * 16)
    y = torch.empty(N, C, HW, dtype=torch.bfloat16, device=dev)
    y3 = torch.full((N, Cy, HW), 7.0, dtype=torch.bfloat16, device=dev)
    a3d, b3d = a3.to(dev), b3.to(dev)
    lib.call("hific_add_split", a3d.data_ptr(), la, b3d.data_ptr(), lb, y.data_ptr(), y3.data_ptr(), lo, N, C, HW, lib.stream())
    torch.cuda.synchronize()
    assert torch.equal(y.cpu(), ref.to(torch.bfloat16))
    assert torch.equal(y3.cpu(), _split_cpu(ref, lo))


def _gen(hific, dev, sd):
    from hific_amd.network.generator import Generator
    gen = Generator((3, 128, 128), 2, C=220, n_residual_blocks=N_RES)
    sub = {k[len("Generator."):]: v for k, v in sd.items() if k.startswith("Generator.")}
    gen.load_state_dict(sub, strict=True)
    return gen.to(dev)


@pytest.fixture(scope="module")
def sd():
    return O.make_state_dict(seed=0, gan=True, n_res=N_RES)


def test_generator_exact_chain_forward_and_gradients(hific, dev, sd):
    from hific_amd import ops
    y = O.make_noise(3, (2, 220, 8, 8)) * 4
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k.startswith("Generator.")}
    yr_in = y.clone().requires_grad_(True)
    xr = O.generator_forward(sdr, yr_in, N_RES)
    g = O.make_noise(4, tuple(xr.shape))
    xr.backward(g)
    ref = {k[len("Generator."):]: v.grad for k, v in sdr.items()}
    ref["input"] = yr_in.grad

    def run(mode):
        ops.set_exact_training(mode != "plain")
        ops.set_exact_generator_fused(mode == "chain")
        gen = _gen(hific, dev, sd)
        yd = y.to(dev).requires_grad_(True)
        x = gen(yd)
        x.backward(g.to(dev).to(x.dtype))
        torch.cuda.synchronize()
        got = {k: p.grad.float().cpu() for k, p in gen.named_parameters()}
        got["input"] = yd.grad.float().cpu()
        return x.detach().float().cpu(), got, x.dtype

    x_plain, g_plain, _ = run("plain")
    x_chain, g_chain, dt = run("chain")
    x_r5, g_r5, _ = run("unfused")
    e_plain, e_chain, e_r5 = _relerr(x_plain, xr.detach()), _relerr(x_chain, xr.detach()), _relerr(x_r5, xr.detach())
    print(f"\n[Generator 2 blocks @128^2] forward max-rel vs oracle: plain bf16 {e_plain:.2e}, exact chain {e_chain:.2e}, "
          f"round-5 exact form {e_r5:.2e}")
    assert dt == torch.float32
    assert e_chain < 2e-4 and e_r5 < 2e-4 and e_plain > 10 * e_chain
    assert set(g_chain) == set(ref)
    check_grads(g_chain, ref, None, 0.6, "Generator exact chain vs oracle (bf16 backward)")
    worst = max(_relerr(g_chain[k], g_plain[k]) for k in ref)
    worst_o = max(_relerr(g_chain[k], ref[k]) for k in ref)
    worst_po = max(_relerr(g_plain[k], ref[k]) for k in ref)
    print(f"  gradients: exact chain vs oracle worst {worst_o:.2e} (plain bf16 vs oracle {worst_po:.2e}); chain vs plain {worst:.2e}")
    # the chain's backward IS the plain bf16 backward at more accurate forward values: it must not be further from the
    # oracle than the plain mode by more than noise
    assert worst_o < max(1.5 * worst_po, 4e-2)


def test_generator_exact_chain_no_grad_route_equals_training_route(hific, dev, sd):
    from hific_amd import ops
    y = (O.make_noise(3, (2, 220, 8, 8)) * 4).to(dev)
    gen = _gen(hific, dev, sd)
    ops.set_exact_training(True)
    a = gen(y).detach()
    ops.set_exact_training(False)
    ops.set_exact_reconstruction(True)
    with torch.no_grad():
        b = gen(y)
    plain = gen(y).detach()                 # grad enabled + reconstruction option only: the plain bf16 forward
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    assert plain.dtype == torch.bfloat16


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_generator_conv_norm_nodes_are_bit_identical_to_separate_ops(hific, dev, sd, dt):
    """ops.ConvNormFn (conv -> ChannelNorm as one autograd node, the plain path's default) launches the same kernels in the
    same order as Conv2dFn / ConvTranspose2dFn + ChannelNormFn: output and every gradient bit for bit."""
    from hific_amd import ops
    hific.set_compute_dtype(dt)
    y = (O.make_noise(3, (2, 220, 8, 8)) * 4)
    g = None
    res = []
    for fused in (True, False):
        ops.set_conv_norm_fused(fused)
        try:
            gen = _gen(hific, dev, sd)
            yd = y.to(dev).requires_grad_(True)
            x = gen(yd)
            if g is None:
                g = O.make_noise(4, tuple(x.shape)).to(dev).to(x.dtype)
            x.backward(g)
            torch.cuda.synchronize()
        finally:
            ops.set_conv_norm_fused(True)
        res.append((x.detach().clone(), yd.grad.clone(), {k: p.grad.clone() for k, p in gen.named_parameters()}))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert set(res[0][2]) == set(res[1][2])
    for k in res[0][2]:
        assert torch.equal(res[0][2][k], res[1][2][k]), k
    hific.set_compute_dtype(torch.float32)


def test_split_in_pack_is_bit_identical_to_the_derived_weight_image(hific, dev, sd):
    """hific_conv2d_fwd flags bit 5 (ops._split_in_pack): the (hi, hi, lo) packed operand of a 3C exact convolution formed by the
    pack pass from the float32 master weight - same bits as hific_split3(which = 1) followed by the ordinary pack: the exact
    Generator chain (960-channel trunk: split-in-pack applies) gives identical outputs and gradients either way, also after
    the weights change (the pack is re-made from the master weight)."""
    from hific_amd import ops
    y = (O.make_noise(3, (2, 220, 8, 8)) * 4)
    ops.set_exact_training(True)
    res = []
    for on in (True, False):
        ops.set_split_in_pack(on)
        ops.pack_cache.clear(); ops.split_weights.clear()
        try:
            gen = _gen(hific, dev, sd)
            outs = []
            for step in range(2):
                yd = y.to(dev).requires_grad_(True)
                x = gen(yd)
                x.sum().backward()
                outs.append((x.detach().clone(), yd.grad.clone()))
                with torch.no_grad():                       # a torch-visible in-place update: the version counters move
                    for p_ in gen.parameters():
                        p_.mul_(1.0 + 1e-3)
            torch.cuda.synchronize()
        finally:
            ops.set_split_in_pack(True)
        res.append(outs)
    for (xa, ga), (xb, gb) in zip(*res):
        assert torch.equal(xa, xb) and torch.equal(ga, gb)
    assert not torch.equal(res[0][0][0], res[0][1][0])       # the update really changed the result

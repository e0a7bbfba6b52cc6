"""Host-side logic that needs no GPU: module construction / state_dict contract, config, schedules, parameter
arenas, gradient-bucket planning."""
import pytest
import torch

from oracle import hific_oracle as O


def test_state_dict_schema_matches_reference_layout(hific):
    from hific_amd.default_config import make_args, hific_args, ModelTypes
    m = hific.Model(make_args(hific_args), model_type=ModelTypes.COMPRESSION_GAN, allow_random_lpips_backbone=True)
    sd = m.state_dict()
    want = {k: tuple(s) for k, s, _ in O._shapes(gan=True)}
    assert set(sd) == set(want) and len(sd) == 168
    for k, v in sd.items():
        assert tuple(v.shape) == want[k], k
    n_amort = sum(p.numel() for mod in m.amortization_models for p in mod.parameters())
    n_hyper = sum(p.numel() for p in m.Hyperprior.hyperlatent_likelihood.parameters())
    n_disc = sum(p.numel() for p in m.Discriminator.parameters())
    assert (n_amort, n_hyper, n_disc) == (181461583, 14080, 2793117)
    # the oracle's fixture state_dict loads unchanged (same keys/shapes as a reference checkpoint)
    m.load_state_dict(O.make_state_dict(seed=1, gan=True), strict=True)
    assert "perceptual_loss" not in "".join(sd)                            # LPIPS is not part of the checkpoint


def test_same_seed_gives_reference_initialisation(hific):
    """Parameters are created by the same torch constructors in the same order as the reference modules, so a
    seed reproduces the reference's initial weights (checked against the reference where it is importable)."""
    from oracle import ref_loader
    from hific_amd.network.encoder import Encoder
    from hific_amd.network.discriminator import Discriminator
    torch.manual_seed(7)
    e1 = Encoder((3, 64, 64), 2, C=8)
    torch.manual_seed(7)
    d1 = Discriminator((3, 64, 64), (8, 4, 4), C=8)
    if not ref_loader.available():
        pytest.skip("reference not present")
    ns = ref_loader.load()
    torch.manual_seed(7)
    e2 = ns.encoder.Encoder((3, 64, 64), 2, C=8)
    torch.manual_seed(7)
    d2 = ns.discriminator.Discriminator((3, 64, 64), (8, 4, 4), C=8)
    for (k1, v1), (k2, v2) in zip(e1.state_dict().items(), e2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2), k1
    s1, s2 = d1.state_dict(), d2.state_dict()
    assert set(s1) == set(s2)
    for k in s1:
        assert torch.equal(s1[k], s2[k]), k


def test_schedules_and_config(hific):
    from hific_amd.helpers.utils import get_scheduled_params
    from hific_amd.default_config import make_args, mse_lpips_args
    sch = dict(vals=[2., 1.], steps=[50000])
    for s in (0, 1, 49999, 50000, 123456):
        assert get_scheduled_params(2.0, sch, s) == O.get_scheduled_params(2.0, sch, s)
    assert get_scheduled_params(2.0, sch, 5, ignore_schedule=True) == 2.0
    a = make_args(mse_lpips_args, regime="high")
    assert a.target_rate == 0.45 and a.lambda_A == 0.5 and a.latent_channels == 220 and a.n_residual_blocks == 9


def test_param_arena_and_adam_bookkeeping(hific):
    from hific_amd import optim
    ps = [torch.nn.Parameter(torch.randn(5, 3)), torch.nn.Parameter(torch.randn(70)), torch.nn.Parameter(torch.randn(1))]
    before = [p.detach().clone() for p in ps]
    arena = optim.ParamArena(ps)
    for p, b in zip(ps, before):
        assert torch.equal(p.detach(), b)
        assert p.data.data_ptr() >= arena.flat.data_ptr()
        assert p.grad is not None and p._hific_slot.grad.data_ptr() == p.grad.data_ptr()
    assert all(o % 64 == 0 for o in arena.offsets)
    s = ps[0]._hific_slot
    assert s.take() == 0 and s.take() == 1          # first write overwrites, second accumulates
    arena.zero_grad()
    assert s.take() == 0
    arena.flat[arena.offsets[1]] = 42.0
    assert ps[1].detach()[0] == 42.0                 # parameters are views of the arena


def test_bucket_planning_reverse_order(hific):
    from hific_amd import optim, parallel
    ps = [torch.nn.Parameter(torch.zeros(n)) for n in (1000, 300000, 50, 9000000, 64, 64)]
    arena = optim.ParamArena(ps)
    red = parallel.BucketedGradReducer(arena, bucket_mbytes=1)
    assert red.world == 1
    covered = sorted((lo, hi) for lo, hi, _ in red.buckets)
    assert covered[0][0] == 0 and covered[-1][1] == arena.numel
    for (a, b), (c, d) in zip(covered, covered[1:]):
        assert b == c
    assert red.buckets[0][1] == arena.numel          # first bucket = tail of the arena (produced first by backward)
    assert sum(n for _, _, n in red.buckets) == len(ps)
    assert red.finish() == 1.0


def test_injection_into_reference_model(hific):
    """The reference's own Model builds the MI355X modules after inject.patch_reference() and accepts a
    reference-layout state_dict."""
    from oracle import ref_loader
    if not ref_loader.available():
        pytest.skip("reference not present")
    import importlib
    import sys
    ns = ref_loader.load()
    import hific_amd.inject as inject
    saved = {}
    names = inject.patch_reference()
    try:
        assert "src.network.encoder.Encoder" in names
        m = ref_loader.build_reference_model(ns, gan=True)
        assert type(m.Encoder).__module__.startswith("hific_amd")
        assert type(m.Generator.resblock_0).__module__.startswith("hific_amd")
        assert type(m.Hyperprior.hyperlatent_likelihood).__module__.startswith("hific_amd")
        assert type(m.Discriminator).__module__.startswith("hific_amd")
        m.load_state_dict(O.make_state_dict(seed=2, gan=True), strict=True)
        assert len(m.state_dict()) == 168
    finally:
        for modname in ("src.network.encoder", "src.network.generator", "src.network.discriminator",
                        "src.network.hyper", "src.normalisation.channel", "src.hyperprior",
                        "src.compression.hyperprior_model", "src.loss.perceptual_similarity.perceptual_loss",
                        "src.model"):
            if modname in sys.modules:
                importlib.reload(sys.modules[modname])


def test_param_arena_rebind_zero_unwritten_and_adam_state_dict(hific):
    """optim.ParamArena survives nn.Module._apply (the reference's save_model does model.cpu() -> .to(device) every
    epoch), does not feed stale gradients to the optimizer, and FusedAdam checkpoints in torch.optim.Adam's format."""
    import torch
    from hific_amd import optim
    torch.manual_seed(0)
    lin = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3))
    params = list(lin.parameters())
    opt = optim.FusedAdam(params, lr=1e-3)
    arena = opt.arena
    assert arena.rebind() == 0
    before = [p.detach().clone() for p in params]
    lin.double().float()                                  # any _apply re-points .data (and .grad) away from the arena
    assert any(p.data_ptr() != arena.flat.data_ptr() + 4 * o for p, o in zip(params, arena.offsets))
    with torch.no_grad():
        params[0].add_(1.0)                               # the user-visible tensor is the truth after a stray
    assert arena.rebind() == len(params)
    for p, o, b in zip(params, arena.offsets, before):
        assert p.data_ptr() == arena.flat.data_ptr() + 4 * o and p.grad.data_ptr() == arena.flat_grad.data_ptr() + 4 * o
    assert torch.equal(params[0], before[0] + 1.0) and torch.equal(params[1], before[1])
    # stale gradients: only slot 2 is written this "backward"; the others must read as zero at step time
    arena.flat_grad.fill_(7.0)
    arena.zero_grad()
    s = params[2]._hific_slot
    assert s.take() == 0
    s.grad.fill_(3.0)
    arena.zero_unwritten()
    assert float(params[2].grad.min()) == 3.0
    for i in (0, 1, 3):
        assert float(params[i].grad.abs().max()) == 0.0
    # state_dict round trip through torch.optim.Adam's own loader
    opt.step_count = 4
    opt.exp_avg.copy_(torch.arange(arena.numel, dtype=torch.float32) * 1e-3)
    opt.exp_avg_sq.copy_(torch.arange(arena.numel, dtype=torch.float32) * 1e-6)
    sd = opt.state_dict()
    ref = torch.optim.Adam([torch.nn.Parameter(p.detach().clone()) for p in params], lr=1.0)
    ref.load_state_dict(sd)                               # torch accepts our format
    assert ref.param_groups[0]["lr"] == 1e-3 and float(ref.state[ref.param_groups[0]["params"][1]]["step"]) == 4.0
    opt2 = optim.FusedAdam([torch.nn.Parameter(p.detach().clone()) for p in params], lr=5.0)
    opt2.load_state_dict(ref.state_dict())                # and we accept torch's
    assert opt2.step_count == 4 and opt2.param_groups[0]["lr"] == 1e-3
    for i in range(len(params)):
        o, n = arena.slice_of(i)
        assert torch.equal(opt2.exp_avg[o:o + n], opt.exp_avg[o:o + n])
        assert torch.equal(opt2.exp_avg_sq[o:o + n], opt.exp_avg_sq[o:o + n])


def test_instance_norm_variant_constructs_like_the_reference(hific):
    """use_channel_norm=False (src/normalisation/instance.py:7-15): documented PyTorch fallback, reference key layout
    (`weight` / `bias` instead of `gamma` / `beta`)."""
    import torch
    from hific_amd.network.encoder import Encoder
    from hific_amd.network.generator import Generator
    from hific_amd.normalisation import instance
    enc = Encoder((3, 64, 64), 2, C=220, channel_norm=False)
    gen = Generator((220, 4, 4), 2, C=220, n_residual_blocks=1, channel_norm=False)
    assert "conv_block1.2.weight" in enc.state_dict() and "conv_block1.2.gamma" not in enc.state_dict()
    assert "resblock_0.norm1.bias" in gen.state_dict() and "upconv_block2.1.weight" in gen.state_dict()
    m = instance.InstanceNorm2D_wrap(5, fuse_relu=True)
    x = torch.randn(2, 5, 6, 7)
    ref = torch.relu(torch.nn.InstanceNorm2d(5, affine=True)(x))
    assert torch.allclose(m(x), ref, atol=1e-6)
    assert m(x.bfloat16()).dtype == torch.bfloat16


def test_option_variants_keep_the_reference_layout(hific):
    """Options that are off in the shipped configs but honoured: Generator(sample_noise=True) has the reference's
    state_dict layout (992-channel blocks; checked against the reference class where the checkout exists), and
    PerceptualLoss(net='vgg') carries VGG16's 13 conv layers with the reference's linear-head widths."""
    import warnings
    from hific_amd.network.generator import Generator
    from hific_amd.loss.perceptual_loss import PerceptualLoss, NETS
    gen = Generator((8, 4, 4), 2, C=8, n_residual_blocks=1, sample_noise=True, noise_dim=32)
    sd = gen.state_dict()
    assert tuple(sd["conv_block_init.2.weight"].shape) == (960, 8, 3, 3)
    assert tuple(sd["resblock_0.conv1.weight"].shape) == (992, 992, 3, 3)
    assert tuple(sd["upconv_block1.0.weight"].shape) == (992, 480, 3, 3)
    from oracle import ref_loader
    if ref_loader.available():
        ns = ref_loader.load()
        ref = ns.generator.Generator((8, 4, 4), 2, C=8, n_residual_blocks=1, sample_noise=True, noise_dim=32)
        rsd = ref.state_dict()
        assert list(rsd.keys()) == list(sd.keys())
        assert all(tuple(rsd[k].shape) == tuple(sd[k].shape) for k in sd)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        pl = PerceptualLoss(net='vgg', use_gpu=False, allow_random_backbone=True)
    assert pl.net == 'vgg' and len(NETS['vgg']) == 13 and sum(1 for c in NETS['vgg'] if c[6]) == 5
    assert [tuple(pl._t[f"lin{i}"].shape) for i in range(5)] == [(64,), (128,), (256,), (512,), (512,)]
    assert tuple(pl._t["features.28.weight"].shape) == (512, 512, 3, 3)
    with pytest.raises(NotImplementedError):
        PerceptualLoss(net='squeeze', use_gpu=False, allow_random_backbone=True)


def test_arena_slots_written_by_autograd_are_tracked(hific):
    """A parameter whose gradient comes from ATen (the InstanceNorm fallback's affine pair) is accumulated in place into
    its arena slot by autograd: the slot must count as written (zero_unwritten keeps it, buckets see it exactly once) and the
    first accumulation after zero_grad() must land on zeros, not on the previous step's values.  Slots
    written by kernels (their Functions return None to autograd) must NOT be touched by this mechanism."""
    import torch
    from hific_amd import optim
    from hific_amd.normalisation import instance
    torch.manual_seed(0)
    norm = instance.InstanceNorm2D_wrap(6, fuse_relu=True)
    kernel_written = torch.nn.Parameter(torch.ones(6))

    class KernelLike(torch.autograd.Function):          # what ops.Conv2dFn does: writes the slot itself, returns None
        @staticmethod
        def forward(ctx, x, w):
            ctx.slot = w._hific_slot
            return x * 1.0

        @staticmethod
        def backward(ctx, g):
            acc = ctx.slot.take()
            val = g.sum(dim=(0, 2, 3))
            ctx.slot.grad.copy_(ctx.slot.grad + val if acc else val)
            ctx.slot.written()
            return g, None

    arena = optim.ParamArena(list(norm.parameters()) + [kernel_written])
    seen = []
    arena.on_write = lambda slot: seen.append(slot.index)
    x = torch.randn(2, 6, 5, 5)

    def grads():
        ref = torch.nn.InstanceNorm2d(6, affine=True)
        ref.load_state_dict({"weight": norm.weight.detach().clone(), "bias": norm.bias.detach().clone()})
        torch.relu(ref(x)).square().sum().backward()
        return ref.weight.grad, ref.bias.grad

    for step in range(3):
        seen.clear()
        KernelLike.apply(norm(x), kernel_written).square().sum().backward()
        gw, gb = grads()
        assert all(not s.fresh for s in arena.slots)
        assert torch.allclose(arena.slots[0].grad, gw, atol=1e-4) and torch.allclose(arena.slots[1].grad, gb, atol=1e-4)
        assert norm.weight.grad.data_ptr() == arena.flat_grad.data_ptr()
        assert sorted(seen) == [0, 1, 2], seen          # each slot reported exactly once per backward
        arena.zero_unwritten()                       # what FusedAdam.step does first: must not wipe these
        assert torch.allclose(arena.slots[0].grad, gw, atol=1e-4)
        arena.zero_grad()
        assert all(s.fresh for s in arena.slots)


def test_pack_cache_unpin_releases_only_the_entries_that_graph_pinned(hific):
    """graph A pins the cache, the cache is cleared (A is invalidated), the same keys are re-created and
    pinned by graph B; closing / collecting A must not release B's pins (it used to identify entries by key)."""
    from hific_amd import ops

    def entry():
        e = ops._PackEntry()
        e.pinned, e.last_use = False, 0
        return e
    pc = ops.WeightPackCache()
    pc.entries = {("k", i): entry() for i in range(3)}
    broken_a, pins_a = pc.pin_all()                       # graph A
    assert all(e.pinned == 1 for e in pc.entries.values())
    pc.clear()                                            # what makes the user re-capture
    assert pc.pins_broken == broken_a + 1
    pc.entries = {("k", i): entry() for i in range(3)}    # same keys, fresh entries
    broken_b, pins_b = pc.pin_all()                       # graph B
    pc.unpin(pins_a)                                      # `del A` -> GraphedStep.close()
    assert all(e.pinned == 1 for e in pc.entries.values()), "graph A's unpin released graph B's entries"
    assert pc.pins_broken == broken_b                     # B stays replayable
    pc.unpin(pins_b)
    assert all(e.pinned == 0 for e in pc.entries.values())
    # two graphs sharing live entries: reference-counted
    _, p1 = pc.pin_all(); _, p2 = pc.pin_all()
    pc.unpin(p1)
    assert all(e.pinned == 1 for e in pc.entries.values())
    pc.unpin(p2)
    assert not any(e.pinned for e in pc.entries.values())


def test_reducer_bucket_layout_with_a_split_tail(hific):
    """BucketedGradReducer cuts the arena from its END in bucket_mbytes slices (backward order) and splits the slice that holds
    slot 0 - the one sealed last, whose all-reduce nothing can hide - by tail_mbytes caps counted from slot 0 upward.  Every
    layout is a partition of the arena into contiguous slot ranges; deferred arenas are not split."""
    import torch
    from hific_amd import optim, parallel
    sizes = (300, 70000, 5, 130000, 64, 9000, 200000)
    ps = [torch.nn.Parameter(torch.zeros(n)) for n in sizes]
    arena = optim.ParamArena(ps)

    def layout(**kw):
        red = parallel.BucketedGradReducer(arena, **kw)
        cover = 0
        for b, (lo, hi, n) in enumerate(red.buckets):
            slots = [i for i in range(len(sizes)) if red.slot_bucket[i] == b]
            assert len(slots) == n and slots == list(range(slots[0], slots[0] + n)) and arena.offsets[slots[0]] == lo
            cover += hi - lo
        assert cover == arena.numel
        return [[i for i in range(len(sizes)) if red.slot_bucket[i] == b] for b in range(len(red.buckets))]

    assert layout(bucket_mbytes=2.0, tail_mbytes=()) == [[0, 1, 2, 3, 4, 5, 6]]
    assert layout(bucket_mbytes=1.0, tail_mbytes=()) == [[4, 5, 6], [0, 1, 2, 3]]
    # caps 2 KiB | 0.3 MiB from slot 0: [0] | [1, 2] | rest of that slice
    assert layout(bucket_mbytes=1.0, tail_mbytes=(0.002, 0.3)) == [[4, 5, 6], [3], [1, 2], [0]]
    # a cap the whole slice fits under splits nothing; a cap smaller than the first slot is skipped
    assert layout(bucket_mbytes=1.0, tail_mbytes=(4.0,)) == [[4, 5, 6], [0, 1, 2, 3]]
    assert layout(bucket_mbytes=1.0, tail_mbytes=(0.0001, 0.3)) == [[4, 5, 6], [3], [0, 1, 2]]
    assert layout(bucket_mbytes=1.0, tail_mbytes=(0.002, 0.3), eager=False) == [[4, 5, 6], [0, 1, 2, 3]]

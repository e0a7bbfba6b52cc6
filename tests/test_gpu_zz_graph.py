"""hipGraph capture of the G-turn + D-turn cycle (hific_amd.graph.GraphedStep): a captured cycle
replayed N times must leave EXACTLY the state N eager cycles leave - parameters of all three optimizer groups, Adam
moments, device-side step counts, spectral-norm buffers, the losses of the last cycle - because the graph holds the same
kernels with the same arguments in the same stream order."""
import itertools

import pytest
import torch

from oracle import hific_oracle as O

pytestmark = pytest.mark.gpu
N_RES, B, S, WARM, REPLAYS = 2, 4, 128, 2, 3


def _setup(dev, device_rng):
    import hific_amd
    from hific_amd import optim, ops
    from hific_amd.default_config import make_args, hific_args, ModelTypes
    hific_amd.set_compute_dtype(torch.bfloat16)
    ops.pack_cache.clear(); ops.split_weights.clear()
    args = make_args(hific_args, n_residual_blocks=N_RES, batch_size=B, image_dims=(3, S, S), latent_dims=(220, S // 16, S // 16))
    model = hific_amd.Model(args, model_type=ModelTypes.COMPRESSION_GAN, device_rate_select=True,
                            allow_random_lpips_backbone=True)
    model.load_state_dict(O.make_state_dict(seed=0, gan=True, n_res=N_RES), strict=True)
    model.perceptual_loss.load_backbone_state_dict(O.make_alex_backbone())
    model = model.to(dev).train()
    amort = [p for m in model.amortization_models for p in m.parameters()]
    opts = {"amort": optim.FusedAdam(amort, lr=1e-4),
            "hyper": optim.FusedAdam(list(model.Hyperprior.hyperlatent_likelihood.parameters()), lr=1e-4),
            "disc": optim.FusedAdam(list(model.Discriminator.parameters()), lr=1e-4)}
    gen = torch.Generator(device=dev).manual_seed(99)
    xs = [O.make_image(s, B, S, S).to(dev) for s in (1, 2)]
    if not device_rng:
        # fixed quantisation noise: two draws per forward, two forwards per cycle, the same every cycle
        noise = [O.make_noise(s, shape).to(dev) for s, shape in
                 ((3, (B, 320, S // 64, S // 64)), (4, (B, 220, S // 16, S // 16)), (5, (B, 320, S // 64, S // 64)),
                  (6, (B, 220, S // 16, S // 16)))]
        it = itertools.cycle(noise)
        model.Hyperprior._draw_noise = lambda t: next(it).clone()

    def batch(i):
        return torch.rand((B, 3, S, S), generator=gen, device=dev) if device_rng else xs[i]

    def step():
        losses = model(batch(0), train_generator=True, writeout=False)
        losses["compression"].backward()
        for n in ("amort", "hyper"):
            opts[n].step(); opts[n].zero_grad()
        g_loss = losses["compression"].detach()
        losses = model(batch(1), train_generator=False, writeout=False)
        losses["disc"].backward()
        opts["disc"].step(); opts["disc"].zero_grad()
        opts["amort"].zero_grad(); opts["hyper"].zero_grad()
        return g_loss, losses["disc"].detach()

    return model, opts, step, gen


def _state(model, opts, out):
    torch.cuda.synchronize()
    st = {f"{k}.params": o.arena.flat.clone() for k, o in opts.items()}
    st.update({f"{k}.m": o.exp_avg.clone() for k, o in opts.items()})
    st.update({f"{k}.v": o.exp_avg_sq.clone() for k, o in opts.items()})
    st.update({f"{k}.step": torch.tensor(o.step_count) for k, o in opts.items()})
    st.update({k: v.clone() for k, v in model.state_dict().items() if "weight_u" in k or "weight_v" in k})
    st["loss_G"], st["loss_D"] = out[0].clone(), out[1].clone()
    return st


def _run(dev, use_graph, device_rng):
    from hific_amd.graph import GraphedStep
    torch.manual_seed(1234)
    model, opts, step, gen = _setup(dev, device_rng)
    if use_graph:
        gs = GraphedStep(step, warmup=WARM, generators=(gen,))
        for _ in range(REPLAYS):
            out = gs()
    else:
        for _ in range(WARM + REPLAYS):
            out = step()
    return _state(model, opts, out)


def test_captured_cycle_is_bit_identical_to_eager(hific, dev):
    eager = _run(dev, False, False)
    graph = _run(dev, True, False)
    assert int(eager["amort.step"]) == WARM + REPLAYS == int(graph["amort.step"]) == int(graph["disc.step"])
    bad = [k for k in eager if not torch.equal(eager[k], graph[k])]
    assert not bad, bad
    assert torch.isfinite(graph["loss_G"]) and torch.isfinite(graph["loss_D"])
    # the replays really trained: parameters differ from a run with fewer cycles
    assert float(graph["loss_G"]) != 0.0


def test_captured_cycle_with_device_rng(hific, dev):
    """Quantisation noise and input batches drawn on the device inside the captured cycle (what bench.py does): torch's
    generators advance their philox offset per replay, so graph and eager consume the same random stream."""
    eager = _run(dev, False, True)
    graph = _run(dev, True, True)
    same = [k for k in eager if torch.equal(eager[k], graph[k])]
    print(f"  device-RNG cycle: {len(same)} of {len(eager)} state tensors bit-identical to eager; "
          f"loss_G eager {float(eager['loss_G']):.6f} graph {float(graph['loss_G']):.6f}")
    assert int(graph["amort.step"]) == WARM + REPLAYS
    assert abs(float(eager["loss_G"]) - float(graph["loss_G"])) < 2e-2 * abs(float(eager["loss_G"]))
    assert all(torch.isfinite(v.float()).all() for v in graph.values())


def test_captured_single_stream_cycle_and_pinned_pack_cache(hific, dev):
    """(i) The cycle captured with everything on ONE stream (no side / branch streams: the only form whose hipGraphLaunch
    takes ROCm's packet-replay path, 0.3 ms of host time per cycle instead of 19 ms) leaves the eager multi-stream state bit
    for bit.  (ii) ADVICE round 3: a captured graph reads the packed-weight cache by raw pointer and never touches its LRU
    bookkeeping - entries that exist at capture are pinned: eager work that floods the cache afterwards must not evict
    them, replacing an entry must keep the byte count exact, and dropping a pinned entry makes the graph refuse to replay."""
    from hific_amd import ops
    from hific_amd.graph import GraphedStep
    eager = _run(dev, False, False)
    side_was, branch_was = ops._SIDE_ON, ops.branch_streams_on()
    ops.set_side_stream(False); ops.set_branch_streams(False)
    try:
        torch.manual_seed(1234)
        model, opts, step, gen = _setup(dev, False)
        gs = GraphedStep(step, warmup=WARM, generators=(gen,))
        out = gs()
        pc = ops.pack_cache
        assert pc.entries and all(e.pinned for e in pc.entries.values())
        pinned = set(pc.entries)
        # flood: a tiny cap and new geometry keys (another batch size through one layer) - the LRU may only take the new ones
        cap_was = pc.cap_bytes
        pc.cap_bytes = 1
        try:
            conv = model.Generator.resblock_0.conv1
            for n in (1, 2, 3):
                with torch.no_grad():
                    conv(torch.zeros(n, 960, 8, 8, device=dev, dtype=torch.bfloat16))
        finally:
            pc.cap_bytes = cap_was
        assert pinned <= set(pc.entries)
        assert pc.bytes == sum(e.buf.numel() for e in pc.entries.values())
        for _ in range(REPLAYS - 1):
            out = gs()
        graph = _state(model, opts, out)
        bad = [k for k in eager if not torch.equal(eager[k], graph[k])]
        assert not bad, bad
        # a pinned entry goes away -> the graph must not replay over freed memory
        pc._drop(next(iter(pinned)))
        with pytest.raises(RuntimeError, match="capture the step again"):
            gs()
    finally:
        ops.set_side_stream(side_was); ops.set_branch_streams(branch_was)
        ops.pack_cache.clear(); ops.split_weights.clear()

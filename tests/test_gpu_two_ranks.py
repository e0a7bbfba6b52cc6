"""Data-parallel semantics of the REAL model: two ranks x local batch 2 through
`hific_amd.Model` + `parallel.BucketedGradReducer` (eager buckets for the amortisation arena, deferred ones for the
hyperprior-density / Discriminator arenas, the global-mean q_bpp rate branch) must give the gradients of

  * compression model: ONE rank with the global batch 4 - every loss is a batch mean and `_estimate_entropy` divides by the
    local batch size (src/hyperprior.py:80-93), so mean-over-ranks of local means = the global mean;
  * compression_gan model: the mean of the two shards' gradients computed one after the other in one process (the
    Discriminator's latent-pairing quirk, src/model.py:176-179, pairs images with latents WITHIN a rank's batch, so a
    sharded batch is not a re-ordering of the global one: SURVEY section 8e).

The reference itself has no multi-GPU path (train.py:303-308 raises).  With >= 2 visible GPUs the ranks use the nccl
(= RCCL) backend, one GPU each; on a one-GPU box both ranks share cuda:0 and reduce through gloo's device-tensor
all-reduce, which exercises the same reducer code (reduce stream, sealing, finish) - only the transport differs."""
import os
import sys

import pytest
import torch

from oracle import hific_oracle as O
from gradcheck import check_grads

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_RES, S, B_LOCAL, WORLD = 2, 128, 2, 2


def _inputs():
    Bg = B_LOCAL * WORLD
    return (O.make_image(5, Bg, S, S), O.make_noise(6, (Bg, 320, S // 64, S // 64)),
            O.make_noise(7, (Bg, 220, S // 16, S // 16)))


def _build(dev, gan, batch):
    import hific_amd
    from hific_amd import optim
    from hific_amd.default_config import make_args, mse_lpips_args, hific_args, ModelTypes
    hific_amd.set_compute_dtype(torch.float32)
    args = make_args(hific_args if gan else mse_lpips_args, n_residual_blocks=N_RES, batch_size=batch,
                     image_dims=(3, S, S), latent_dims=(220, S // 16, S // 16))
    model = hific_amd.Model(args, model_type=ModelTypes.COMPRESSION_GAN if gan else ModelTypes.COMPRESSION,
                            allow_random_lpips_backbone=True)
    sd = O.make_state_dict(seed=0, gan=gan, n_res=N_RES)
    model.load_state_dict(sd, strict=True)
    model.perceptual_loss.load_backbone_state_dict(O.make_alex_backbone())
    model = model.to(dev).train()
    amort = [p for m in model.amortization_models for p in m.parameters()]
    hyper = list(model.Hyperprior.hyperlatent_likelihood.parameters())
    arenas = {"amort": optim.ParamArena(amort), "hyper": optim.ParamArena(hyper)}
    if gan:
        arenas["disc"] = optim.ParamArena(list(model.Discriminator.parameters()))
    return model, sd, arenas


def _g_turn(model, x, nh, nl, dev):
    noises = [nh.to(dev), nl.to(dev)]
    model.Hyperprior._draw_noise = lambda t: noises.pop(0)
    losses = model(x.to(dev), train_generator=True, writeout=False)
    losses["compression"].backward()
    return float(losses["compression"])


def _rank_main(rank, world, port, gan, backend, outfile, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          HIFIC_BUCKET_MB="4", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        import torch.distributed as dist
        local = rank if backend == "nccl" else 0
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        dist.init_process_group(backend, rank=rank, world_size=world)
        try:
            probe = torch.full((8,), float(rank + 1), device=dev)
            try:
                dist.all_reduce(probe)
                torch.cuda.synchronize()
                ok = bool((probe == 3.0).all())
            except Exception as e:                       # this torch build's gloo cannot reduce device tensors
                q.put((rank, f"skip: {backend} all_reduce on device tensors unavailable ({type(e).__name__})"))
                return
            if not ok:
                q.put((rank, "skip: device all_reduce returned wrong values")); return
            from hific_amd import parallel
            model, sd, arenas = _build(dev, gan, B_LOCAL)
            red = {k: parallel.BucketedGradReducer(a, eager=(k == "amort")) for k, a in arenas.items()}
            assert red["amort"].world == world and red["amort"].active and len(red["amort"].buckets) >= 3
            x, nh, nl = _inputs()
            sl = slice(rank * B_LOCAL, (rank + 1) * B_LOCAL)
            loss = _g_turn(model, x[sl], nh[sl], nl[sl], dev)
            scales = {k: r.finish() for k, r in red.items()}
            torch.cuda.synchronize()
            assert all(v == 1.0 / world for v in scales.values())
            grads = {k: (p.grad.detach().float() * scales["amort"]).cpu() for k, p in model.named_parameters()}
            # every rank must now hold the same reduced gradient
            chk = torch.stack([g.double().sum() for g in grads.values()]).to(dev)
            lo, hi = chk.clone(), chk.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            assert torch.equal(lo, hi), "ranks disagree on the reduced gradients"
            lt = torch.tensor([loss], device=dev, dtype=torch.float64)
            dist.all_reduce(lt)
            if rank == 0:
                torch.save({"grads": grads, "mean_loss": float(lt.item()) / world}, outfile)
            q.put((rank, "ok"))
        finally:
            dist.destroy_process_group()
    except Exception as e:  # noqa
        import traceback
        q.put((rank, "error: " + repr(e) + "\n" + traceback.format_exc()[-1500:]))


def _run_ranks(gan, tmp_path):
    import torch.multiprocessing as mp
    backend = "nccl" if torch.cuda.device_count() >= WORLD else "gloo"
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    outfile = str(tmp_path / "rank0.pt")
    port = 29600 + (os.getpid() % 300) + (7 if gan else 0)
    procs = [ctx.Process(target=_rank_main, args=(r, WORLD, port, gan, backend, outfile, q)) for r in range(WORLD)]
    for p in procs:
        p.start()
    res = {}
    try:
        for _ in range(WORLD):
            r, msg = q.get(timeout=420)
            res[r] = msg
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.terminate()
    if any(m.startswith("skip") for m in res.values()):
        pytest.skip(next(m for m in res.values() if m.startswith("skip")))
    assert all(m == "ok" for m in res.values()), res
    print(f"  2 ranks over {backend}: ok")
    return torch.load(outfile, weights_only=False)


def _compare(got, want, tol, what):
    """Same rule as the oracle comparisons (tests/gradcheck.py): per-tensor max error <= tol of the tensor's scale; a tensor
    beyond it is accepted only with the signature of a ReLU sign tie (the planner picks kernels by grid size, so batch 2 and
    batch 4 forwards differ by float32 summation order, and a pre-activation at +-1e-7 can get the other mask): the
    beyond-bar elements sit in <= 8 channel slices."""
    check_grads(got, want, None, tol, what)


def test_two_ranks_equal_one_rank_with_the_global_batch(hific, dev, tmp_path):
    sharded = _run_ranks(False, tmp_path)
    model, _, arenas = _build(dev, False, B_LOCAL * WORLD)
    x, nh, nl = _inputs()
    loss = _g_turn(model, x, nh, nl, dev)
    torch.cuda.synchronize()
    want = {k: p.grad.detach().float().cpu() for k, p in model.named_parameters()}
    assert abs(sharded["mean_loss"] - loss) < 1e-5 * abs(loss), (sharded["mean_loss"], loss)
    # float32 on both sides; the planner picks kernels by grid size, so batch 2 and batch 4 differ in summation order (and in
    # the ReLU sign ties that follow from it): the same 1e-3 bar as against the oracle
    _compare(sharded["grads"], want, 1e-3, "compression model: 2 ranks x 2 vs 1 rank x 4")


def test_two_ranks_gan_equal_the_mean_of_the_shards(hific, dev, tmp_path):
    sharded = _run_ranks(True, tmp_path)
    x, nh, nl = _inputs()
    acc, losses = None, []
    for r in range(WORLD):
        model, _, arenas = _build(dev, True, B_LOCAL)          # fresh spectral-norm buffers, like every rank starts with
        sl = slice(r * B_LOCAL, (r + 1) * B_LOCAL)
        losses.append(_g_turn(model, x[sl], nh[sl], nl[sl], dev))
        torch.cuda.synchronize()
        g = {k: p.grad.detach().float().cpu() / WORLD for k, p in model.named_parameters()}
        acc = g if acc is None else {k: acc[k] + g[k] for k in g}
        del model, arenas
    assert abs(sharded["mean_loss"] - sum(losses) / WORLD) < 1e-5 * abs(losses[0])
    _compare(sharded["grads"], acc, 2e-5, "compression_gan: 2 ranks vs mean of the two shards' gradients")


def test_bench_multi_rank_control_flow_rehearsal():
    """`bench.py` as the driver launches it for N > 1 (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`),
    rehearsed on ONE GPU: HIFIC_BENCH_REHEARSAL=1 puts every rank on cuda:0 with gloo as the transport (RCCL refuses two ranks
    on one device).  Same reducers, same collectives in the same order, same report: a collective that one rank issues and
    another does not (as the rate rule's global q_bpp mean did inside scale_report's rank-0-alone pass) hangs here, under the
    timeout, instead of on the 8-GPU node.  stdout must be the one JSON line, with the multi-rank block filled in."""
    import json
    import subprocess
    env = dict(os.environ, HIFIC_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("HIFIC_FORCE_DIST", None)
    port = 29700 + os.getpid() % 200
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, res.stdout[:2000]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["config"]["parallelism"] == "dp2" and d["config"]["global_batch"] == 32
    r = d["rccl"]
    assert r["rccl_ranks"] == 2 and "REHEARSAL" in r["backend"]
    assert set(r["payload_sweep_ms_per_step"]) == {"f32", "bf16"}
    assert r["one_rank_same_box_ms_per_step"] > 0 and r["weak_scaling_eff"] > 0
    assert all(len(v) >= 1 for v in r["buckets_timeline"].values())

"""world_size-2 data-parallel path on CPU (gloo): bucketed gradient reduction over a ParamArena with the eager
(overlapped) and deferred modes, and the scalar mean used for the rate-penalty branch."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, eager, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from hific_amd import optim, parallel
        torch.manual_seed(0)
        ps = [torch.nn.Parameter(torch.randn(n)) for n in (300, 70000, 5, 130000, 64)]
        arena = optim.ParamArena(ps)
        red = parallel.BucketedGradReducer(arena, bucket_mbytes=0.25, eager=eager)
        assert red.world == world and len(red.buckets) >= 3
        red.measure_exposed(True); red.measure_timeline(True)        # the measurement hooks of bench.py's multi-rank line must not disturb the reduction
        for it in range(2):                                  # two backward passes: bookkeeping must reset
            for i in reversed(range(len(ps))):               # backward order
                s = ps[i]._hific_slot
                acc = s.take()
                val = float(rank + 1) * (i + 1) + it
                if acc:
                    s.grad.add_(val)
                else:
                    s.grad.fill_(val)
                s.written()
            scale = red.finish()
            assert scale == 1.0 / world
            for i, p in enumerate(ps):
                want = sum(float(r + 1) * (i + 1) + it for r in range(world))
                assert torch.allclose(p.grad, torch.full_like(p.grad, want)), (it, i)
            arena.zero_grad()
        assert red.bucket_timeline() == [] and red.exposed_comm_ms() == 0.0     # device events only: nothing on CPU tensors
        m = parallel.allreduce_scalar_mean(torch.tensor(float(rank)))
        assert abs(float(m) - (world - 1) / 2.0) < 1e-6
        q.put((rank, "ok"))
    except Exception as e:  # noqa
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def _worker_multiwrite(rank, world, port, q):
    """A training-step-shaped write pattern: parameter 1 is written TWICE per backward (a module applied twice per
    forward, e.g. HyperpriorDensity with both likelihood evaluations in the loss).  With its expected write count
    registered the bucket is reduced only after the second accumulation; unregistered, the second write is refused
    loudly instead of racing the in-flight all-reduce."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from hific_amd import optim, parallel
        ps = [torch.nn.Parameter(torch.zeros(n)) for n in (300, 70000, 5, 130000)]
        arena = optim.ParamArena(ps)
        red = parallel.BucketedGradReducer(arena, bucket_mbytes=0.25, eager=True, expected_writes={ps[1]: 2})

        def write(i, val):
            s = ps[i]._hific_slot
            if s.take():
                s.grad.add_(val)
            else:
                s.grad.fill_(val)
            s.written()

        for it in range(2):
            write(3, 1.0 + rank)
            write(1, 10.0 * (rank + 1))          # first use of the shared module
            assert not red.launched[red.slot_bucket[1]], "bucket sealed before the slot's last write"
            write(2, 2.0)
            write(1, 5.0)                         # second use: accumulates, THEN the bucket may go
            write(0, float(it))
            assert red.finish() == 0.5
            assert torch.allclose(ps[1].grad, torch.full_like(ps[1].grad, 10.0 * 3 + 5.0 * 2))
            assert torch.allclose(ps[3].grad, torch.full_like(ps[3].grad, 3.0))
            assert torch.allclose(ps[2].grad, torch.full_like(ps[2].grad, 4.0))
            arena.zero_grad()
        # the same pattern WITHOUT the registration must fail loudly on the late write
        red2 = parallel.BucketedGradReducer(arena, bucket_mbytes=0.25, eager=True)
        write(3, 1.0); write(1, 1.0)
        try:
            write(1, 1.0)
            q.put((rank, "late write was accepted")); return
        except RuntimeError as e:
            assert "expected_writes" in str(e)
        red2.finish()
        # global-batch rate branch: both ranks pick the same lambda from the mean q_bpp
        from hific_amd.loss import losses
        from hific_amd.default_config import make_args, mse_lpips_args
        a = make_args(mse_lpips_args)
        target = a.target_rate * a.target_schedule["vals"][0]
        qbpp = torch.tensor(target + (0.3 if rank == 0 else -0.1))       # rank 0 above, rank 1 below, mean above
        _, pen = losses.weighted_rate_loss(a, torch.tensor(1.0), qbpp, step_counter=1)
        _, pen_dev = losses.weighted_rate_loss(a, torch.tensor(1.0), qbpp, step_counter=1, device_select=True)
        assert pen == a.lambda_A * a.lambda_schedule["vals"][0] and float(pen_dev) == pen, (pen, float(pen_dev))
        q.put((rank, "ok"))
    except Exception as e:  # noqa
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_multiwrite_slots_and_global_rate_branch_world2(hific):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29850 + (os.getpid() % 100)
    procs = [ctx.Process(target=_worker_multiwrite, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(30)
    assert all(r[1] == "ok" for r in res), res


def _worker_bf16_payload(rank, world, port, q):
    """payload="bf16": the wire carries bfloat16, the arena gets float32 sums back; element-wise
    |result - sum_r g_r| <= 2^-7 sum_r |g_r| (one rounding per term, one per addition), and exactly-representable gradients
    (small integers) come back exact."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from hific_amd import optim, parallel
        sizes = (300, 70000, 5, 130000, 64)
        ps = [torch.nn.Parameter(torch.zeros(n)) for n in sizes]
        arena = optim.ParamArena(ps)
        for eager in (True, False):
            red = parallel.BucketedGradReducer(arena, bucket_mbytes=0.25, eager=eager, payload="bf16")
            assert red.payload == "bf16" and len(red.buckets) >= 3
            for it in range(2):
                gs = {r: [torch.randn(n, generator=torch.Generator().manual_seed(1000 * it + 10 * r + i)) * 10.0 ** (i - 2)
                          for i, n in enumerate(sizes)] for r in range(world)}
                for i in reversed(range(len(ps))):
                    s = ps[i]._hific_slot
                    assert s.take() == 0
                    s.grad.copy_(gs[rank][i] if it == 0 else torch.full((sizes[i],), float(rank + 2 + i)))
                    s.written()
                assert red.finish() == 1.0 / world
                for i, p in enumerate(ps):
                    assert p.grad.dtype == torch.float32
                    if it == 0:
                        exact = sum(gs[r][i].double() for r in range(world))
                        bound = 2.0 ** -7 * sum(gs[r][i].abs().double() for r in range(world)) + 1e-30
                        assert bool(((p.grad.double() - exact).abs() <= bound).all()), (eager, i)
                        assert float((p.grad.double() - exact).abs().max()) > 0.0 or sizes[i] < 10   # it WAS rounded
                    else:
                        assert torch.equal(p.grad, torch.full_like(p.grad, float(sum(r + 2 + i for r in range(world)))))
                arena.zero_grad()
            arena.on_write = None
        q.put((rank, "ok"))
    except Exception:  # noqa
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_bf16_gradient_payload_world2(hific):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29950 + (os.getpid() % 40)
    procs = [ctx.Process(target=_worker_bf16_payload, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(30)
    assert all(r[1] == "ok" for r in res), res


@pytest.mark.parametrize("eager", [True, False])
def test_bucketed_allreduce_world2(hific, eager):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 200) + (1 if eager else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, eager, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(30)
    assert all(r[1] == "ok" for r in res), res


def _worker_solo_rank(rank, world, port, q):
    """bench.py scale_report: rank 0 steps ALONE (reducers off) while the other ranks wait in a barrier.  With the scalar
    collectives off, rank 0's allreduce_scalar_mean calls issue nothing, so the next collective every rank takes part in (the
    barrier) still pairs up; back on, the global mean is the global mean again."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from hific_amd import parallel
        if rank == 0:
            was = parallel.set_scalar_collectives(False)
            assert was is True
            for _ in range(5):
                t = torch.tensor(3.0)
                assert float(parallel.allreduce_scalar_mean(t)) == 3.0
            assert parallel.set_scalar_collectives(was) is False
        dist.barrier()
        m = parallel.allreduce_scalar_mean(torch.tensor(float(rank + 1)))
        assert abs(float(m) - (world + 1) / 2.0) < 1e-6
        q.put((rank, "ok"))
    except Exception as e:  # noqa
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_one_rank_can_step_alone_without_issuing_collectives_world2(hific):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29870 + (os.getpid() % 40)
    procs = [ctx.Process(target=_worker_solo_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(30)
    assert all(r[1] == "ok" for r in res), res


def test_tail_split_buckets_reduce_to_the_same_sums_world2(hific, monkeypatch):
    """The last-sealed bucket split into [first layers | next | rest] ($HIFIC_BUCKET_TAIL_MB): a different cut of the same arena,
    the same sums.  1 MiB buckets over (300, 70000, 5, 130000, 64) floats: [3, 4] | [1, 2] | [0]."""
    monkeypatch.setenv("HIFIC_BUCKET_MB", "1.0")
    monkeypatch.setenv("HIFIC_BUCKET_TAIL_MB", "0.002,0.3")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29810 + (os.getpid() % 40)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, True, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(30)
    assert all(r[1] == "ok" for r in res), res

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
        m = parallel.allreduce_scalar_mean(torch.tensor(float(rank)))
        assert abs(float(m) - (world - 1) / 2.0) < 1e-6
        q.put((rank, "ok"))
    except Exception as e:  # noqa
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


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

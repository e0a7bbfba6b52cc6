"""Which ATen ops / memcpys (copies, fills, elementwise glue) still launch device work inside one training cycle, with their
Python call sites:        python tools/trace_aten.py            (GPU box)"""
import os
import sys
import argparse
import collections

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402

args = argparse.Namespace(gpus=1, steps=2, warmup=2, batch=16, size=256, config="gan", dtype="bf16", regime="low", seed=0)
dev = torch.device("cuda:0")
model, opts, reducers = bench.build(args, dev, "gan")
step = bench.make_step(args, model, opts, reducers, dev, "gan")
for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity  # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step()
    torch.cuda.synchronize()


def dev_time(e):
    for a in ("self_device_time_total", "self_cuda_time_total"):
        if hasattr(e, a):
            return getattr(e, a)
    return 0


agg = collections.OrderedDict()
for e in prof.events():
    if dev_time(e) <= 0 or not (e.name.startswith("aten::") or "emcpy" in e.name or "emset" in e.name):
        continue
    st = [s for s in (e.stack or []) if ("/repo/" in s or "hific" in s) and "trace_aten" not in s][:3]
    key = (e.name, " <- ".join(s.split("repo/")[-1][:80] for s in st), str(getattr(e, "input_shapes", ""))[:60])
    a = agg.setdefault(key, [0, 0.0])
    a[0] += 1; a[1] += dev_time(e)
tot_n = sum(a[0] for a in agg.values()); tot_t = sum(a[1] for a in agg.values())
print(f"=== ATen / memcpy events with device time in ONE cycle: {tot_n} launches, {tot_t / 1e3:.3f} ms ===")
for (name, stack, shapes), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print(f"{name:22s} x{n:3d} {t:8.1f} us  {shapes:60s} {stack}")

"""Soak: N eager GAN cycles on one GPU; device memory (allocated / reserved), packed-weight cache size and the losses must
be flat / finite.   python tools/soak.py [cycles]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from hific_amd import ops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
args = argparse.Namespace(batch=16, size=256, dtype="bf16", regime="low", seed=0, steps=8, warmup=3)
dev = torch.device("cuda:0")
model, opts, reducers = bench.build(args, dev, "gan")
step = bench.make_step(args, model, opts, reducers, dev, "gan")
marks = {}
t0 = time.perf_counter()
for i in range(1, n + 1):
    out = step()
    if i in (20, n // 2, n):
        torch.cuda.synchronize()
        marks[i] = (torch.cuda.memory_allocated(dev), torch.cuda.memory_reserved(dev), len(ops.pack_cache.entries))
        vals = {k: round(float(v.detach()), 5) for k, v in out.items() if torch.is_tensor(v) and v.numel() == 1}
        print(f"[soak] cycle {i}: allocated {marks[i][0] / 2**20:.1f} MiB, reserved {marks[i][1] / 2**20:.1f} MiB, "
              f"pack-cache entries {marks[i][2]}, outputs {vals}, {(time.perf_counter() - t0) / i * 1e3:.2f} ms/cycle", flush=True)
ks = sorted(marks)
assert marks[ks[-1]][0] <= marks[ks[0]][0] * 1.01 + (1 << 20), "allocated memory grows"
# (the caching allocator's reserve settles during the first few dozen cycles - streams hand blocks over with a delay - so the
# reserve is compared between the middle and the end)
assert marks[ks[-1]][1] <= marks[ks[1]][1] * 1.01 + (1 << 20), "reserved memory grows"
for p in model.parameters():
    assert torch.isfinite(p).all()
print("[soak] ok")

"""Which ATen ops (copies, fills, elementwise glue) still run inside one training cycle, with shapes and Python call sites:
    python tools/trace_aten.py            (GPU box)"""
import os
import sys
import argparse

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
print(prof.key_averages(group_by_input_shape=True).table(sort_by="count", row_limit=60, max_name_column_width=40,
                                                         max_shapes_column_width=60))
print("=== by stack (copy_/clone/fill_/add/mul) ===")
for e in prof.key_averages(group_by_stack_n=6):
    if e.key in ("aten::copy_", "aten::clone", "aten::fill_", "aten::zero_", "aten::add", "aten::mul", "aten::to", "aten::_to_copy",
                 "aten::contiguous", "aten::cat", "aten::where", "aten::uniform_", "aten::zeros_like", "aten::empty_like") and e.count >= 2:
        st = [s for s in e.stack if "repo" in s or "hific" in s][:3]
        print(f"{e.key:18s} x{e.count:4d}  " + " <- ".join(s.split("repo/")[-1][:70] for s in st))

"""A/B of the training cycle: plain bf16 | exact training as the fused Generator chain (round 6) | round-5 exact form.
    python tools/r06/exact_ab.py [steps]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from hific_amd import ops  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
args = argparse.Namespace(gpus=1, batch=16, size=256, dtype="bf16", regime="low", seed=0, steps=steps, warmup=3, config="gan")
dev = torch.device("cuda:0")


def fence():
    torch.cuda.synchronize()


def leg(name, exact, fused):
    ops.set_exact_training(exact)
    ops.set_exact_generator_fused(fused)
    ops.pack_cache.clear(); ops.split_weights.clear()
    model, opts, reducers = bench.build(args, dev, "gan")
    step = bench.make_step(args, model, opts, reducers, dev, "gan")
    for _ in range(3):
        step()
    e = bench.timed(step, steps, 1, fence)
    print(f"[exact_ab] {name}: {e / steps * 1e3:.2f} ms per cycle = {32 * steps / e:.1f} images/s", flush=True)
    del model, opts, reducers, step
    torch.cuda.empty_cache()


for rep in range(2):
    leg("plain bf16", False, True)
    leg("exact training, fused chain", True, True)
    leg("exact training, round-5 form", True, False)
ops.set_exact_training(False)

#!/usr/bin/env python
"""HiFIC hot-path benchmark (contract: see the task brief / DESIGN.md §Measurement).

  python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

A "step" = one training step of the HiFIC-low *compression* model (BASELINE.json configs[1]): forward
(Encoder -> Hyperprior -> Generator -> MSE + LPIPS + rate) + backward + the two Adam updates (amortisation and
hyperprior-density parameters, reference train.py:54-59), bf16 compute with f32 master weights, batch 16 of
synthetic 256x256 RGB crops per GPU, random-init weights.  `--config gan` times configs[2] (G-turn + D-turn).
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# algorithmic work per image of the compression training step: 151.9 GMAC = 303.7 GFLOP
FLOP_PER_IMAGE_COMPRESSION = 303.7e9
MFMA_PEAK_BF16_TFLOPS = 2500.0   # dense, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16, help="images per GPU")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--config", default="compression", choices=["compression", "gan"])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=4)
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--cpu-timeout", type=int, default=150)
    return ap.parse_args()


def build(args, dev):
    import hific_amd
    from hific_amd import optim, parallel
    from hific_amd.default_config import make_args, mse_lpips_args, hific_args, ModelTypes
    hific_amd.set_compute_dtype(torch.bfloat16 if args.dtype == "bf16" else torch.float32)
    gan = args.config == "gan"
    torch.manual_seed(0)
    margs = make_args(hific_args if gan else mse_lpips_args, batch_size=args.batch,
                      image_dims=(3, args.size, args.size), latent_dims=(220, args.size // 16, args.size // 16))
    model = hific_amd.Model(margs, model_type=ModelTypes.COMPRESSION_GAN if gan else ModelTypes.COMPRESSION,
                            device_rate_select=True)
    model = model.to(dev).train()
    # parameter groups exactly as train.py:287-301
    amort = []
    for m in model.amortization_models:
        amort += list(m.parameters())
    hyper = list(model.Hyperprior.hyperlatent_likelihood.parameters())
    opts = {"amort": optim.FusedAdam(amort, lr=1e-4), "hyper": optim.FusedAdam(hyper, lr=1e-4)}
    if gan:
        opts["disc"] = optim.FusedAdam(list(model.Discriminator.parameters()), lr=1e-4)
    reducers = {k: parallel.BucketedGradReducer(o.arena, eager=(k != "disc")) for k, o in opts.items()}
    return model, opts, reducers


def make_step(args, model, opts, reducers, dev):
    gan = args.config == "gan"
    gen = torch.Generator(device=dev).manual_seed(1234 + int(os.environ.get("RANK", "0")))
    B, S = args.batch, args.size

    def batch():
        return torch.rand((B, 3, S, S), generator=gen, device=dev, dtype=torch.float32)

    def reduce_and_step(names):
        for n in names:
            opts[n].grad_scale = reducers[n].finish()
            opts[n].step()
            opts[n].zero_grad()

    def step():
        # G-turn (train.py:119-127 / 139-141)
        losses = model(batch(), train_generator=True, writeout=False)
        losses["compression"].backward()
        reduce_and_step(["amort", "hyper"])
        if gan:
            # D-turn on a distinct batch (train.py:129-136); the D gradients deposited by the G-turn stay in the
            # slots (reference quirk): disc.zero_grad() only runs after disc.step()
            losses = model(batch(), train_generator=False, writeout=False)
            losses["disc"].backward()
            reduce_and_step(["disc"])
            opts["amort"].zero_grad(); opts["hyper"].zero_grad()
        return losses

    return step


def _cpu_baseline_worker(size, B, steps, threads):
    """Runs in a child process (no GPU context): reference arithmetic (oracle restatement, torch CPU float32)."""
    import numpy as np
    from oracle import hific_oracle as O
    torch.set_num_threads(threads)
    sd = {k: torch.nn.Parameter(v) for k, v in O.make_state_dict(seed=0, gan=False).items()}
    bb = O.make_alex_backbone()
    lpips_w = np.load(os.path.join(ROOT, "high-fidelity-generative-compression_amd", "loss", "weights",
                                   "lpips_alex_lin_v0.1.npz"))
    lins = [torch.from_numpy(lpips_w[f"lin{i}"].copy()) for i in range(5)]
    hyper_keys = [k for k in sd if "hyperlatent_likelihood" in k]
    amort = torch.optim.Adam([v for k, v in sd.items() if k not in hyper_keys], lr=1e-4)
    hyper = torch.optim.Adam([sd[k] for k in hyper_keys], lr=1e-4)
    times = []
    for it in range(steps + 1):
        x = O.make_image(100 + it, B, size, size)
        t0 = time.time()
        out = O.model_forward(sd, bb, lins, x, step_counter=it + 1, training=True, gan=False)
        out["compression"].backward()
        amort.step(); hyper.step()
        amort.zero_grad(); hyper.zero_grad()
        times.append(time.time() - t0)
    t = sorted(times[1:])[len(times[1:]) // 2]
    print(json.dumps({"images_per_s": B / t, "step_s": t}), flush=True)


def cpu_baseline(args):
    """The oracle ("port" of the reference's CPU path) timed on this box's host cores on a bounded sample of the
    same workload: compression training step (fwd + bwd + 2x Adam), batch `cpu_batch`, 1 warm-up + `cpu_steps`
    timed steps, in a child process with a hard time limit so the GPU line can never be lost to it."""
    import subprocess
    ncores = os.cpu_count() or 1
    threads = min(ncores, int(os.environ.get("HIFIC_CPU_THREADS", "32")))
    cmd = [sys.executable, "-c",
           f"import sys; sys.path.insert(0, {ROOT!r}); import bench; "
           f"bench._cpu_baseline_worker({args.size}, {args.cpu_batch}, {args.cpu_steps}, {threads})"]
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", OMP_NUM_THREADS=str(threads))
    sample = (f"oracle (torch-CPU float32 restatement of the reference) compression train step, batch "
              f"{args.cpu_batch}, {args.size}x{args.size}, 1 warm-up + {args.cpu_steps} timed steps (median), "
              f"{threads} threads of {ncores} logical cores")
    try:
        res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=args.cpu_timeout)
        line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
        r = json.loads(line)
        return {"value": round(r["images_per_s"], 4), "unit": "images/s", "cores": threads, "kind": "port",
                "sample": sample}
    except Exception as e:
        return {"value": None, "unit": "images/s", "cores": threads, "kind": "port",
                "sample": sample + f" -- FAILED: {type(e).__name__}"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or os.environ.get("HIFIC_FORCE_DIST") == "1"   # the latter: 1-rank RCCL smoke test
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=rank, world_size=world)
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    from hific_amd import lib
    model, opts, reducers = build(args, dev)
    step = make_step(args, model, opts, reducers, dev)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # live roofline of the dominant kernel: same steps again with HIP events around every GEMM-class launch
    lib.call("hific_prof_begin")
    for _ in range(max(1, min(args.steps, 4))):
        step()
    ms = (ctypes.c_double * 4)(); fl = (ctypes.c_double * 4)(); cnt = (ctypes.c_int * 4)()
    lib.call("hific_prof_end", ms, fl, cnt)
    kinds = ["gconv 128-row tiles (gconv_sp9_kernel<2> + gconv_kernel<..,2,2,2,2>)",
             "gconv 64-row tiles (gconv_sp9_kernel<1> + gconv_kernel<..,2,2,1,2>)",
             "gconv 32-row tiles (gconv_kernel<..,1,4,1,1>)",
             "weight gradient (wgrad_pipe_kernel + wgrad_kernel + wgrad_im2col_kernel)"]
    kinds_key = ["gconv128", "gconv64", "gconv32", "wgrad"]
    per_kind = {kinds[i]: {"launches": cnt[i], "ms": round(ms[i], 3),
                           "tflops": round(fl[i] / (ms[i] * 1e-3) / 1e12, 2) if ms[i] > 0 else 0.0}
                for i in range(4) if cnt[i] > 0}
    dom = max(range(4), key=lambda i: ms[i])
    traffic = None
    try:   # HBM bytes per launch of the dominant kernel class: rocprofv3 FETCH_SIZE(x2 on gfx950)+WRITE_SIZE, see profiles/
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            traffic = json.load(f).get(kinds_key[dom])
    except Exception:
        traffic = None
    achieved = fl[dom] / (ms[dom] * 1e-3) / 1e12 if ms[dom] > 0 else 0.0
    imgs_per_step = args.batch * (2 if args.config == "gan" else 1)
    value = world * imgs_per_step * args.steps / elapsed
    out = {
        "metric": "training images/sec (256x256) HiFIC-low", "value": round(value, 3), "unit": "images/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"HiFIC-low {'compression_gan (G-turn + D-turn)' if args.config == 'gan' else 'compression'}"
                               f" training step, batch {args.batch}/GPU, {args.size}x{args.size} RGB, {args.dtype} compute, "
                               f"f32 master weights, fwd+bwd+Adam, random-init weights",
                   "global_batch": world * args.batch, "parallelism": f"dp{world}"},
        "roofline": {"bound": "mfma", "kernel": kinds[dom], "achieved": round(achieved, 2),
                     "peak": MFMA_PEAK_BF16_TFLOPS if args.dtype == "bf16" else 157.3, "unit": "TFLOP/s",
                     "frac": round(achieved / (MFMA_PEAK_BF16_TFLOPS if args.dtype == "bf16" else 157.3), 4),
                     "avg_launch_us": round(ms[dom] * 1e3 / max(cnt[dom], 1), 2), "traffic": traffic,
                     "per_kernel": per_kind,
                     "step_model": {"algorithmic_tflop_per_step": round(FLOP_PER_IMAGE_COMPRESSION * args.batch / 1e12, 3),
                                    "whole_step_tflops": round(FLOP_PER_IMAGE_COMPRESSION * args.batch /
                                                               (elapsed / args.steps) / 1e12, 2)
                                    if args.config == "compression" else None}},
    }
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

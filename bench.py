#!/usr/bin/env python
"""HiFIC hot-path benchmark (contract: task brief / DESIGN.md section 5).

  python bench.py --gpus N --steps K --warmup W

N > 1 without a torch.distributed environment re-executes itself under `python -m torch.distributed.run` with N ranks
(one per GPU, RCCL); when the driver launches the ranks itself (RANK / WORLD_SIZE set) it runs as that rank.

Headline (`value`): BASELINE.json configs[2] - HiFIC-low *compression_gan* training, batch 16 of synthetic 256x256
RGB crops per GPU, bf16 MFMA compute with float32 master weights.  A "step" is one G-D cycle as the reference's
train loop runs it (train.py:119-141): G-turn (forward Encoder -> Hyperprior -> Generator -> MSE + LPIPS + rate + D,
backward, Adam on the amortisation and hyperprior-density groups) on one batch, then D-turn (full forward, D loss,
backward, Adam on the Discriminator) on the next batch: 2 x 16 images per step.

The same JSON line also carries (N = 1 only, measured in the same process right after the headline):
  * `compression`: configs[1], the no-GAN model's training step (16 images per step)
  * `fwd_ms_per_image`: EVALUATION-mode forward (model.py:357-366: reconstruction + q_bpp, no losses), no-grad
  * `roofline`: the dominant GEMM kernel FUNCTION of the headline step, timed live with HIP event pairs on the
    launch stream (in-library profiler; single-stream execution for these steps, see profile_kernels), algorithmic FLOPs on the op's real output domain, against the dense bf16
    MFMA peak; `traffic` = HBM bytes per launch of that kernel from two rocprofv3 --pmc passes (FETCH_SIZE,
    WRITE_SIZE) of a short child run of this script, or null
  * `parity`: the benchmarked mode against the oracle, measured in this process at the benchmarked shape (batch 16 x 256^2,
    same weights / images / noise): quantised-index flip rate (+ worst distance of a flipped index from a rounding tie),
    relative error of the loss and of both rates
  * `f32`: the same cycle in float32 parity mode (images/s of the mode whose parity bar is 1e-3 everywhere)
  * `config5_1gpu`: BASELINE configs[4] per-GPU shape (regime high, ONE 1024 x 1024 crop per turn)
  * `cpu_baseline`: the REFERENCE's own modules (unpacked from oracle/_ref/reference_src.tar.gz; kind "reference") or, if
    that archive is missing, the oracle port, on the host cores: bounded sample of the same G-D cycle at BASELINE
    configs[0]'s batch 4, all-core and best thread count, plus the src.model smoke forward (batch 10) the reference
    publishes a time for
With N > 1: `rccl` = ranks, buckets per reducer and `exposed_comm_ms` (GPU time per step the compute stream waits in
BucketedGradReducer.finish() for collectives the backward pass did not hide).
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic work per image, GMAC -> FLOP = 2 x
GFLOP_PER_IMAGE = {
    "compression": 303.7,            # training step of the no-GAN model: fwd 51.68 + bwd 100.18 GMAC
    "forward": 103.4,                # 51.68 GMAC
}
GMAC_G_TURN, GMAC_D_TURN = 161.2, 62.8      # per image of the G-turn / D-turn batch (D weight-gradient quirk included)
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}      # dense, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_BF16_TFLOPS = MFMA_PEAK_TFLOPS["bf16"]
PEAK_HBM_GBPS = 8000.0                                  # HBM3E, same guide


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16, help="images per GPU per turn")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--config", default="gan", choices=["gan", "compression"], help="headline workload")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-extras", action="store_true", help="headline only (no compression/fwd/roofline legs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 --pmc child runs")
    ap.add_argument("--cpu-batch", type=int, default=4)
    ap.add_argument("--cpu-steps", type=int, default=1)
    ap.add_argument("--cpu-timeout", type=int, default=120)
    ap.add_argument("--regime", default="low", choices=["low", "med", "high"])
    ap.add_argument("--no-parity", action="store_true", help="skip the in-process oracle comparison")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--traffic-child", action="store_true", help=argparse.SUPPRESS)
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------------
def build(args, dev, config):
    import torch
    import hific_amd
    from hific_amd import optim, parallel
    from hific_amd.default_config import make_args, mse_lpips_args, hific_args, ModelTypes
    hific_amd.set_compute_dtype(torch.bfloat16 if args.dtype == "bf16" else torch.float32)
    gan = config == "gan"
    torch.manual_seed(args.seed)                 # identical initial weights on every rank
    margs = make_args(hific_args if gan else mse_lpips_args, regime=getattr(args, "regime", "low"), batch_size=args.batch,
                      image_dims=(3, args.size, args.size), latent_dims=(220, args.size // 16, args.size // 16))
    model = hific_amd.Model(margs, model_type=ModelTypes.COMPRESSION_GAN if gan else ModelTypes.COMPRESSION,
                            device_rate_select=True, allow_random_lpips_backbone=True)
    model = model.to(dev).train()
    # parameter groups exactly as train.py:287-301
    amort = [p for m in model.amortization_models for p in m.parameters()]
    hyper = list(model.Hyperprior.hyperlatent_likelihood.parameters())
    # overlap_from: the Encoder's parameters are updated on the main stream, the rest of the group (Generator, hyper
    # nets: 95 % of the bytes) and their weight re-packs on the optimizer stream, under the next step's Encoder forward
    n_enc = len([p for p in model.Encoder.parameters() if p.requires_grad])
    opts = {"amort": optim.FusedAdam(amort, lr=1e-4, overlap_from=n_enc), "hyper": optim.FusedAdam(hyper, lr=1e-4)}
    if gan:
        opts["disc"] = optim.FusedAdam(list(model.Discriminator.parameters()), lr=1e-4)
    # amort: every slot is written exactly once per backward -> buckets go to RCCL as backward produces them;
    # hyper (14 080 parameters, a module applied twice per forward) and disc (written by both turns): one
    # all-reduce in finish()
    reducers = {k: parallel.BucketedGradReducer(o.arena, eager=(k == "amort")) for k, o in opts.items()}
    # weights are built: from here on the default generators (host and device) draw the quantisation noise
    # (src/hyperprior.py:65) - a different stream on every rank (ranks must not share their noise)
    torch.manual_seed(args.seed + 1000003 * (1 + int(os.environ.get("RANK", "0"))))
    return model, opts, reducers


def make_step(args, model, opts, reducers, dev, config):
    import torch
    gan = config == "gan"
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
            # D-turn on the next batch (train.py:129-136); the D gradients deposited by the G-turn stay in the
            # slots (reference quirk): disc.zero_grad() only runs after disc.step()
            losses = model(batch(), train_generator=False, writeout=False)
            losses["disc"].backward()
            reduce_and_step(["disc"])
            opts["amort"].zero_grad(); opts["hyper"].zero_grad()
        return losses

    step.generators = (gen,)
    return step


def graphed(args, step, world, force=False, single_stream=True):
    """The step as one hipGraph replay (hific_amd.graph.GraphedStep).  Round 4 (tools/host_floor_probe.py,
    profiles/r04_host_floor.md): hipGraphLaunch takes ROCm's packet-replay path only for a graph captured on ONE stream -
    0.3 ms of host time per cycle; with the side / branch streams in the capture it walks the ~1000 nodes at 19 us each
    (18.8 ms, what round 3 measured).  So the step is captured single-stream (same kernels, same order: bit-identical,
    tests/test_gpu_zz_graph.py) and choose_launch() below times it against eager multi-stream launching."""
    flag = os.environ.get("HIFIC_BENCH_GRAPH", "auto")
    if flag == "0" and not force:
        return step, False
    from hific_amd import ops
    from hific_amd.graph import GraphedStep
    side_was, branch_was = ops._SIDE_ON, ops.branch_streams_on()
    if single_stream:
        ops.set_side_stream(False)
        ops.set_branch_streams(False)
    try:
        return GraphedStep(step, warmup=max(2, args.warmup), generators=step.generators), True
    except Exception as e:                    # a capture failure must not cost the measurement
        # a failed capture leaves its streams in capture mode for the rest of the process: start over without graphs
        print(f"[bench] hipGraph capture failed ({type(e).__name__}: {str(e).splitlines()[0]}); re-running eagerly",
              file=sys.stderr, flush=True)
        if world > 1:
            raise
        os.environ["HIFIC_BENCH_GRAPH"] = "0"
        os.execv(sys.executable, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:])
    finally:
        ops.set_side_stream(side_was)
        ops.set_branch_streams(branch_was)


def choose_launch(args, step, world, fence, ncal=4):
    """How the timed cycles are launched: eager (one launch per kernel from Python, weight gradients / loss branch on side
    streams) or one single-stream hipGraph replay per cycle.  Both run the same kernels on the same data and leave the same
    state; which is faster depends on whether the host (eager: ~22 ms of enqueueing per cycle) or the serialised kernel
    sum (graph) binds.  `auto` (default at one rank) times `ncal` untimed cycles of each and takes the faster one; the
    calibration is reported in the line.  HIFIC_BENCH_GRAPH=0 / 1 forces eager / graph.  Multi-rank runs launch eagerly:
    the bucketed all-reduce lives on its own stream."""
    flag = os.environ.get("HIFIC_BENCH_GRAPH", "auto")
    # (HIFIC_FORCE_DIST: the one-rank RCCL smoke run drives the reducer - reduce stream, timed events - like a multi-rank job)
    if world > 1 or flag == "0" or os.environ.get("HIFIC_FORCE_DIST") == "1":
        return step, False, None
    gstep, ok = graphed(args, step, world, force=True)
    if not ok:
        return step, False, None
    cal = {}
    for name, fn in (("eager_multi_stream", step), ("graph_single_stream", gstep)):
        fn(); fence(); t0 = time.perf_counter()
        for _ in range(ncal):
            fn()
        th = time.perf_counter() - t0
        fence(); te = time.perf_counter() - t0
        cal[name] = {"ms_per_step": round(te / ncal * 1e3, 3), "host_ms_per_step": round(th / ncal * 1e3, 3)}
    use_graph = flag == "1" or cal["graph_single_stream"]["ms_per_step"] <= cal["eager_multi_stream"]["ms_per_step"]
    cal["chosen"] = "graph_single_stream" if use_graph else "eager_multi_stream"
    cal["how"] = f"{ncal} untimed cycles of each launch mode after the warm-up, same model state; faster one runs the timed region"
    return (gstep if use_graph else step), use_graph, cal


def scale_report(args, step, reducers, opts, world, rank, dev, fence, ms_step):
    """What a multi-rank line needs to explain itself (rank 0 reports; every rank takes part): the per-bucket timeline of the
    last backward (issue and duration on the reduce stream), the same cycle with the other gradient payload, and the same
    cycle on rank 0 ALONE (reducers off, the other ranks parked in a barrier) - `weak_scaling_eff` = that time / the job's."""
    import torch
    import torch.distributed as dist
    from hific_amd import parallel
    n = max(3, min(args.steps, 6))
    for r in reducers.values():
        r.measure_exposed(False)
        r.measure_timeline(True)            # a separate short pass: the per-bucket events serialise the reduce stream
    step(); step()
    torch.cuda.synchronize()
    rep = {"buckets_timeline": {k: [{"bucket": b, "wire_mbytes": mb, "issue_ms": ti, "duration_ms": td}
                                    for b, mb, ti, td in r.bucket_timeline()] for k, r in reducers.items()},
           "buckets_timeline_note": "one extra backward after the timed region with per-bucket events on (they make the reduce "
                                    "stream wait for each collective - not on while timing): issue = ms after the first bucket "
                                    "reached its collective on the reduce stream, duration = collective start to end"}
    for r in reducers.values():
        r.measure_timeline(False)
    # ---- the other payload, same model state ------------------------------------------------------------------------------
    cur = next(iter(reducers.values())).payload
    other = "bf16" if cur == "f32" else "f32"
    sweep = {cur: round(ms_step, 3)}
    for r in reducers.values():
        r.payload = other
    step(); step()
    e = timed(step, n, 0, fence)
    t = torch.tensor([e], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    sweep[other] = round(float(t.item()) / n * 1e3, 3)
    for r in reducers.values():
        r.payload = cur
    rep["payload_sweep_ms_per_step"] = sweep
    # ---- rank 0 alone on the same box: no collectives, the other ranks idle in a barrier -------------------------------------
    solo = None
    was = {k: r.active for k, r in reducers.items()}
    if rank == 0:
        for k, r in reducers.items():
            r.active = False
            r.arena.on_write = None
        # the model's own collective (global mean of q_bpp for the rate rule, two per cycle) has no partner either
        scalars_were = parallel.set_scalar_collectives(False)
        step(); step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        solo = (time.perf_counter() - t0) / n * 1e3
        parallel.set_scalar_collectives(scalars_were)
        for k, r in reducers.items():
            r.active = was[k]
            if r.active and r.eager:
                r.arena.on_write = r._on_write
    fence()
    t = torch.tensor([solo or 0.0], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    solo = float(t.item())
    rep["one_rank_same_box_ms_per_step"] = round(solo, 3)
    rep["weak_scaling_eff"] = round(solo / ms_step, 4) if ms_step > 0 else None
    rep["weak_scaling_note"] = ("rank 0 alone on this box (reducers off, other ranks idle in a barrier; note that the solo "
                                "model's parameters then drift from the other ranks': measurement runs only) over the "
                                f"{world}-rank step time")
    return rep


def timed(step, steps, warmup, fence):
    for _ in range(warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    t_enq = time.perf_counter() - t0            # host finished enqueueing (diagnostic: host-bound if ~ the total)
    fence()
    dt = time.perf_counter() - t0
    if os.environ.get("HIFIC_BENCH_DIAG"):
        print(f"[bench diag] {steps} steps: host enqueue {t_enq * 1e3 / steps:.2f} ms/step, total {dt * 1e3 / steps:.2f} ms/step",
              file=sys.stderr)
    return dt


def profile_kernels(step, nsteps):
    """HIP-event pairs around every GEMM-class launch of `nsteps` further steps -> per kernel function totals."""
    from hific_amd import lib, ops
    MAXK = 48
    # Kernel durations are taken with everything on ONE stream: in the timed headline the weight gradients run on a side
    # stream (ops._SideLaunch) concurrently with the data-gradient chain, which would charge each kernel for the time it
    # shares the chip with another one; likewise the loss branch / Discriminator branch streams of model.py.  (Same setting in
    # the rocprofv3 passes under profiles/: HIFIC_SIDE_WGRAD=0 HIFIC_BRANCH_STREAMS=0.)
    side_was, branch_was = ops._SIDE_ON, ops.branch_streams_on()
    ops.set_side_stream(False)
    ops.set_branch_streams(False)
    lib.call("hific_prof_begin")
    try:
        for _ in range(nsteps):
            step()
    finally:
        ops.set_side_stream(side_was)
        ops.set_branch_streams(branch_was)
    ms = (ctypes.c_double * MAXK)(); fl = (ctypes.c_double * MAXK)(); cnt = (ctypes.c_int * MAXK)()
    names = ctypes.create_string_buffer(MAXK * 64)
    by = (ctypes.c_double * MAXK)()
    lib.raw("hific_prof_bytes")(MAXK, by)
    nk = lib.raw("hific_prof_end")(MAXK, ms, fl, cnt, names)
    if nk < 0:
        raise RuntimeError(f"hific_prof_end failed ({nk})")
    out = {}
    for k in range(nk):
        nm = names.raw[k * 64:(k + 1) * 64].split(b"\0", 1)[0].decode()
        if cnt[k] > 0:
            out[nm] = {"launches_per_step": cnt[k] / nsteps, "ms_per_step": ms[k] / nsteps,
                       "avg_launch_us": ms[k] * 1e3 / cnt[k], "gflop_per_launch": fl[k] / cnt[k] / 1e9,
                       "tflops": fl[k] / (ms[k] * 1e-3) / 1e12 if ms[k] > 0 else 0.0}
            # Kernel functions whose launches sit below the machine balance (2500 TFLOP/s / 8 TB/s = 312 FLOP per byte of
            # operands read once + result written once) are priced against the HBM roofline: the few-channel layers on
            # 256 x 256 planes, the first / last convolutions' weight gradients.
            if by[k] > 0 and ms[k] > 0 and fl[k] / by[k] < PEAK_BF16_TFLOPS * 1e12 / (PEAK_HBM_GBPS * 1e9):
                gbps = by[k] / (ms[k] * 1e-3) / 1e9
                out[nm]["hbm_bound"] = {"achieved_GBps": gbps, "peak_GBps": PEAK_HBM_GBPS, "frac": gbps / PEAK_HBM_GBPS,
                                        "algorithmic_MB_per_launch": by[k] / cnt[k] / 1e6,
                                        "flop_per_byte": fl[k] / by[k]}
    return out


# ---- HBM traffic of one kernel function: rocprofv3 --pmc child runs ---------------------------------------------
def _rocpd_counter(dbdir, counter, kernel_substr, grid=None):
    """-> (average counter value per launch of the kernels matching `kernel_substr` (compared without spaces; `grid` = total
    work-items of the launch, when given), sum over ALL launches)."""
    import sqlite3
    tot, n, everything = 0.0, 0, 0.0
    want = kernel_substr.replace(" ", "")
    for db in glob.glob(os.path.join(dbdir, "**", "*.db"), recursive=True):
        cur = sqlite3.connect(db).cursor()
        try:
            rows = cur.execute("select kernel_name, grid_size, count(*), sum(value) from counters_collection "
                               "where counter_name=? group by kernel_name, grid_size", (counter,)).fetchall()
        except Exception:
            continue
        for name, g, c, v in rows:
            everything += v
            if want in name.replace(" ", "") and (grid is None or g == grid):
                tot += v; n += c
    return ((tot / n) if n else None), everything


def _trace_filter(kind):
    """In-library profiler kind -> (substring of the rocprofv3 kernel name, total work-items or None).  The residual-block trunk
    launches of gconv_sp9_kernel<2,4> (the profiler's own class: K, C >= 512) are the 256-workgroup grids of the non-gather
    instantiation; the 220 / 320-channel launches of the same function have 320 / 768 workgroups."""
    if kind == "gconv_sp9_kernel<2,4>":
        return "gconv_sp9_kernel<2,4,false", 256 * 512
    return kind, None


def practical_peak():
    """The measured ceiling of the dominant kernel's tile: the residual-trunk convolution (960 -> 960, 3x3, 16 x 16x16, the
    same launch geometry) run from libhific_hip_mfma_only.so - the library built with gconv_sp9_kernel's patch loads, weight
    loads, LDS fragment reads and chunk barriers compiled out (csrc/build.sh, -DSP9_ABL=15; results are wrong by
    construction), i.e. the bare v_mfma_f32_32x32x16_bf16 stream of that tile on register-resident non-zero operands.  Timed by
    the in-library HIP events in a child process (tools/micro_sp9.py).  `roofline.peak` stays the data-sheet 2.5 PF."""
    so = os.path.join(ROOT, "high-fidelity-generative-compression_amd", "libhific_hip_mfma_only.so")
    tool = os.path.join(ROOT, "tools", "micro_sp9.py")
    if not (os.path.exists(so) and os.path.exists(tool)):
        return None
    env = dict(os.environ, HIFIC_LIB_PATH=so, MOPS="fwd", HIFIC_TICKETS="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    try:
        out = subprocess.run([sys.executable, tool, "40"], env=env, capture_output=True, text=True, timeout=120).stdout
    except Exception:
        return None
    for line in out.splitlines():
        if "gconv_sp9_kernel<2,4>" in line and "narrow" not in line and " us x" in line:
            try:
                us = float(line.split(":")[-1].split("us")[0])
                tf = float(line.rsplit("x", 1)[-1].split()[1])
            except Exception:
                continue
            return {"tflops": tf, "avg_launch_us": us, "unit": "TFLOP/s",
                    "how": "the trunk launch (67.95 GFLOP) with every memory instruction of gconv_sp9_kernel<2,4> compiled out "
                           "(libhific_hip_mfma_only.so, -DSP9_ABL=15): the MFMA issue rate of this tile on resident, non-zero "
                           "operands on THIS box - the ceiling a perfect operand pipeline would reach"}
    return None


def measure_traffic(args, kernel_name):
    """Two separate counter passes (FETCH_SIZE / WRITE_SIZE cannot share one: TCC slots), each a short child run
    of this script (1 warm-up + 1 step, headline only).  Units: KB; FETCH_SIZE counts 64 B per 128-B request of a
    wide stream on gfx950 -> read bytes = 2 x FETCH_SIZE (MI355X_MICROARCH.md, HBM section)."""
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None, "rocprofv3 not found"
    sub = kernel_name.split("<")[0] if "gconv_kernel" not in kernel_name else "gconv_kernel"
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="hific_pmc_", dir="/tmp")
        cmd = [rocprof, "--pmc", counter, "-d", d, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__),
               "--traffic-child", "--steps", "3", "--warmup", "1", "--batch", str(args.batch), "--size", str(args.size),
               "--config", args.config, "--dtype", args.dtype]
        # counters are per dispatch: single-stream execution in the child, like the timing leg (profile_kernels)
        env = dict(os.environ, TMPDIR="/tmp", HIFIC_SIDE_WGRAD="0", HIFIC_BRANCH_STREAMS="0")
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
            env.pop(k, None)
        try:
            p = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                                 start_new_session=True)
            try:
                p.wait(timeout=150)
            except subprocess.TimeoutExpired:
                os.killpg(p.pid, 9)
                return None, f"{counter} pass timed out"
            # exact template instance when it is unambiguous in the trace, else the function family
            v, everything = _rocpd_counter(d, counter, *_trace_filter(kernel_name))
            if v is None:
                v, everything = _rocpd_counter(d, counter, sub)
        finally:
            shutil.rmtree(d, ignore_errors=True)
        if v is None:
            return None, f"{counter}: kernel not found in the counter database"
        vals[counter] = v
        vals[counter + "_all"] = everything
    rd, wr = 2.0 * vals["FETCH_SIZE"] * 1024, vals["WRITE_SIZE"] * 1024
    # every kernel of the child's 4 cycles (1 warm-up + 3): whole-step HBM traffic against the SURVEY section 8(d) estimate
    whole = (2.0 * vals["FETCH_SIZE_all"] + vals["WRITE_SIZE_all"]) * 1024 / 4.0
    alg = (32 * 0.275 + 6.9) * 1e9 * (args.batch / 16.0) * (args.size / 256.0) ** 2 if args.config == "gan" else \
        (16 * 0.216 + 6.9) * 1e9 * (args.batch / 16.0) * (args.size / 256.0) ** 2
    return {"bytes_per_launch": round(rd + wr), "read_bytes": round(rd), "write_bytes": round(wr),
            "whole_step_hbm_gbytes": round(whole / 1e9, 2), "whole_step_algorithmic_gbytes": round(alg / 1e9, 2),
            "whole_step_traffic_ratio": round(whole / alg, 2),
            "how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), KB x 1024, FETCH x 2 (gfx950); whole "
                   "step = all kernels of 4 cycles / 4; algorithmic = activations 5 passes x 2 B + 38 B per parameter "
                   "(SURVEY section 8d)"}, None


# ---- CPU baseline -----------------------------------------------------------------------------------------------
def _cpu_baseline_worker(size, B, steps, threads, gan):
    """Runs in a child process (no GPU context): reference arithmetic (oracle restatement, torch CPU float32)."""
    import numpy as np
    import torch
    from oracle import hific_oracle as O
    torch.set_num_threads(threads)
    sd = {k: (torch.nn.Parameter(v) if v.dtype.is_floating_point and "weight_u" not in k and "weight_v" not in k else v)
          for k, v in O.make_state_dict(seed=0, gan=gan).items()}
    bb = O.make_alex_backbone()
    lpips_w = np.load(os.path.join(ROOT, "high-fidelity-generative-compression_amd", "loss", "weights",
                                   "lpips_alex_lin_v0.1.npz"))
    lins = [torch.from_numpy(lpips_w[f"lin{i}"].copy()) for i in range(5)]
    hyper_keys = [k for k in sd if "hyperlatent_likelihood" in k]
    disc_keys = [k for k in sd if k.startswith("Discriminator.") and isinstance(sd[k], torch.nn.Parameter)]
    amort_keys = [k for k in sd if isinstance(sd[k], torch.nn.Parameter) and k not in hyper_keys and k not in disc_keys]
    amort = torch.optim.Adam([sd[k] for k in amort_keys], lr=1e-4)
    hyper = torch.optim.Adam([sd[k] for k in hyper_keys], lr=1e-4)
    disc = torch.optim.Adam([sd[k] for k in disc_keys], lr=1e-4) if gan else None
    times = []
    for it in range(steps + 1):
        xa, xb = O.make_image(100 + 2 * it, B, size, size), O.make_image(101 + 2 * it, B, size, size)
        t0 = time.time()
        out = O.model_forward(sd, bb, lins, xa, step_counter=it + 1, training=True, gan=gan, train_generator=True)
        out["compression"].backward()
        amort.step(); hyper.step()
        amort.zero_grad(); hyper.zero_grad()
        if gan:
            for k, v in out["new_uv"].items():
                sd[k] = v
            out = O.model_forward(sd, bb, lins, xb, step_counter=it + 1, training=True, gan=True, train_generator=False)
            out["disc"].backward()
            disc.step(); disc.zero_grad(); amort.zero_grad(); hyper.zero_grad()
            for k, v in out["new_uv"].items():
                sd[k] = v
        times.append(time.time() - t0)
    t = sorted(times[1:])[len(times[1:]) // 2]
    print(json.dumps({"images_per_s": B * (2 if gan else 1) / t, "step_s": t}), flush=True)


REF_TAR = os.path.join(ROOT, "oracle", "_ref", "reference_src.tar.gz")


def _cpu_reference_worker(size, B, steps, thread_counts, gan, fwd_batch):
    """Child process, no GPU: the REFERENCE's own modules (src.model.Model and everything under it, unpacked from
    oracle/_ref/reference_src.tar.gz) driven the way train.py:119-141 drives them - G-turn (backward, Adam on the
    amortisation + hyperprior-density groups), then D-turn on the next batch (Adam on the Discriminator), the three
    torch.optim.Adam of train.py:287-301 - for each thread count; then the src.model smoke forward (model.py:443-463)."""
    import tarfile
    import numpy as np
    import torch
    from oracle import ref_loader
    root = tempfile.mkdtemp(prefix="hific_ref_", dir="/tmp")
    with tarfile.open(REF_TAR) as tar:
        tar.extractall(root)
    # the archive holds sources only: the LPIPS linear heads (6 KB, v0.1) are written from the package's copy of them
    wdir = os.path.join(root, "src", "loss", "perceptual_similarity", "weights", "v0.1")
    os.makedirs(wdir, exist_ok=True)
    lw = np.load(os.path.join(ROOT, "high-fidelity-generative-compression_amd", "loss", "weights", "lpips_alex_lin_v0.1.npz"))
    torch.save({f"lin{i}.model.1.weight": torch.from_numpy(lw[f"lin{i}"].copy()).reshape(1, -1, 1, 1) for i in range(5)},
               os.path.join(wdir, "alex.pth"))
    ns = ref_loader.load(root)
    torch.manual_seed(0)
    model = ref_loader.build_reference_model(ns, gan=gan, batch_size=B, image_dims=(3, size, size),
                                             latent_dims=(220, size // 16, size // 16))
    amort = [p for m in model.amortization_models for p in m.parameters()]
    hyper = list(model.Hyperprior.hyperlatent_likelihood.parameters())
    opt_a, opt_h = torch.optim.Adam(amort, lr=1e-4), torch.optim.Adam(hyper, lr=1e-4)
    opt_d = torch.optim.Adam(model.Discriminator.parameters(), lr=1e-4) if gan else None
    g = torch.Generator().manual_seed(1)

    def cycle():
        x = torch.rand((B, 3, size, size), generator=g)
        losses = model(x, train_generator=True)
        losses["compression"].backward()
        opt_a.step(); opt_h.step(); opt_a.zero_grad(); opt_h.zero_grad()
        if gan:
            x = torch.rand((B, 3, size, size), generator=g)
            losses = model(x, train_generator=False)
            losses["disc"].backward()
            opt_d.step(); opt_d.zero_grad()

    res = {"per_threads": {}}

    def emit():                                              # a line per stage: the parent keeps the last one it got
        best = max(res["per_threads"], key=lambda k: res["per_threads"][k])
        res["best_threads"], res["images_per_s"] = int(best), res["per_threads"][best]
        print(json.dumps(res), flush=True)

    samples = {}
    for i, th in enumerate(thread_counts):
        torch.set_num_threads(th)
        if i == 0:
            cycle()                                          # warm-up (allocator, oneDNN primitives) at the first count only
        ts = []
        for _ in range(steps):
            t0 = time.time(); cycle(); ts.append(time.time() - t0)
        samples[th] = ts
        t = sorted(ts)[len(ts) // 2]
        res["per_threads"][str(th)] = round(B * (2 if gan else 1) / t, 4)
        emit()
        if i == 0 and fwd_batch:
            x = torch.randn((fwd_batch, 3, size, size), generator=g)
            with torch.no_grad():
                model(x)
                t0 = time.time(); model(x); res["fwd_s"], res["fwd_threads"] = round(time.time() - t0, 3), th
            emit()
    # at least three timed samples at the best thread count; the value is their median
    # (the leader can change as its median settles: keep sampling whichever count leads until the leader has three)
    while True:
        best = int(max(res["per_threads"], key=lambda k: res["per_threads"][k]))
        ts = samples[best]
        if len(ts) >= 3:
            break
        torch.set_num_threads(best)
        t0 = time.time(); cycle(); ts.append(time.time() - t0)
        t = sorted(ts)[len(ts) // 2] if len(ts) % 2 else 0.5 * (sorted(ts)[len(ts) // 2 - 1] + sorted(ts)[len(ts) // 2])
        res["per_threads"][str(best)] = round(B * (2 if gan else 1) / t, 4)
    res["samples_at_best"] = [round(B * (2 if gan else 1) / v, 4) for v in samples[best]]
    emit()
    shutil.rmtree(root, ignore_errors=True)


def cpu_baseline(args):
    """the reference modules on this box's host cores - in a child process with a hard time limit so
    the GPU line can never be lost to it.  kind "reference" (the reference's own src.model.Model, from oracle/_ref) or, when
    that archive did not travel, "port" (the oracle restatement).  Bounded sample: batch 4 (BASELINE configs[0]), one
    warm-up + `--cpu-steps` timed G-D cycles at 32 / 64 / physical-core threads, then 3 samples at the best count (median)."""
    ncores = os.cpu_count() or 1
    gan = args.config == "gan"
    what = "compression_gan G-turn + D-turn cycle" if gan else "compression training step"
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    env.pop("OMP_NUM_THREADS", None)
    if os.path.exists(REF_TAR) and os.environ.get("HIFIC_CPU_BASELINE", "reference") == "reference":
        # the usual optimum (32) first, then half and ALL logical cores (SURVEY section 8d asks for the latter; on the 256-thread
        # EPYC box 128 threads are 3x SLOWER than 32 and one 256-thread cycle does not finish in 3 minutes - torch's CPU
        # kernels oversubscribe - so whatever the time limit cuts off is reported as absent)
        # 32 (the usual optimum), 64, and one thread per PHYSICAL core.  All logical cores (256 on the EPYC 9575F box) never
        # finished one cycle inside the time limit in rounds 2-3 (torch's CPU kernels oversubscribe) and is not attempted.
        try:
            import psutil
            phys = psutil.cpu_count(logical=False) or max(1, ncores // 2)
        except Exception:
            phys = max(1, ncores // 2)
        ths = sorted({min(ncores, 32), min(ncores, 64), min(ncores, phys)})
        cmd = [sys.executable, "-c",
               f"import sys; sys.path.insert(0, {ROOT!r}); import bench; "
               f"bench._cpu_reference_worker({args.size}, {args.cpu_batch}, {args.cpu_steps}, {ths}, {gan}, 10)"]
        sample = (f"reference modules (src.model.Model from oracle/_ref, torch CPU float32) {what} as train.py:119-141 runs "
                  f"it (fwd + bwd + 3x torch.optim.Adam), batch {args.cpu_batch} per turn (BASELINE configs[0] batch), "
                  f"{args.size}x{args.size}, 1 warm-up, then {args.cpu_steps} timed at each of {ths} threads ({phys} physical / "
                  f"{ncores} logical cores), then the best count re-timed to 3 samples; value = their median")
        try:
            try:
                stdout = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=args.cpu_timeout).stdout
            except subprocess.TimeoutExpired as te:           # keep the stages that finished
                stdout = te.stdout.decode() if isinstance(te.stdout, bytes) else (te.stdout or "")
                sample += f" -- stopped at the {args.cpu_timeout} s limit, thread counts not reached are absent"
            r = json.loads([l for l in stdout.splitlines() if l.startswith("{")][-1])
            out = {"value": r["images_per_s"], "unit": "images/s", "cores": r["best_threads"], "kind": "reference",
                   "sample": sample, "images_per_s_by_threads": r["per_threads"],
                   "samples_at_best_images_per_s": r.get("samples_at_best"),
                   "physical_core_images_per_s": r["per_threads"].get(str(min(ncores, phys)))}
            if "fwd_s" in r:
                out["src_model_forward_b10_s"] = r["fwd_s"]
                out["src_model_forward_b10_threads"] = r.get("fwd_threads")
                out["src_model_forward_b10_note"] = ("src/model.py:443-463 smoke forward (GAN model, batch 10 x 256^2, both "
                                                     "losses); the reference publishes ~45 s on a 2.8 GHz Core i7 "
                                                     "(src/README.md:112)")
            return out
        except Exception as e:
            why = f"reference worker failed ({type(e).__name__}); "
    else:
        why = ""
    threads = min(ncores, int(os.environ.get("HIFIC_CPU_THREADS", "32")))
    cmd = [sys.executable, "-c",
           f"import sys; sys.path.insert(0, {ROOT!r}); import bench; "
           f"bench._cpu_baseline_worker({args.size}, {args.cpu_batch}, {args.cpu_steps}, {threads}, {gan})"]
    env["OMP_NUM_THREADS"] = str(threads)
    sample = (f"{why}oracle (torch-CPU float32 restatement of the reference) {what} (fwd + bwd + Adam), batch "
              f"{args.cpu_batch} per turn, {args.size}x{args.size}, 1 warm-up + {args.cpu_steps} timed (median), "
              f"{threads} threads of {ncores} logical cores")
    try:
        res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=args.cpu_timeout)
        r = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
        return {"value": round(r["images_per_s"], 4), "unit": "images/s", "cores": threads, "kind": "port",
                "sample": sample}
    except Exception as e:
        return {"value": None, "unit": "images/s", "cores": threads, "kind": "port",
                "sample": sample + f" -- FAILED: {type(e).__name__}"}


# ---- Model.compress / Model.decompress at ~1 megapixel -------------------------------------------
def codec_leg(args, dev):
    """EVALUATION path end to end: x -> Encoder -> hyperprior nets -> symbols (device, csrc/entropy.hip) -> rANS coder
    (host C++, csrc/host_rans.cpp) and back through the Generator.  One 1024 x 1024 image (the reference quotes "2-3
    seconds" for decoding ~megapixel images on a GPU *without* running the rANS coder, src/README.md:87).  Host time is the
    time spent inside compression.rans.ans_compress / ans_decompress (wrapped here), device time the rest."""
    import torch
    import hific_amd
    from hific_amd.compression import rans, codec
    from hific_amd.default_config import make_args, hific_args, ModelTypes, ModelModes
    S = 1024
    hific_amd.set_compute_dtype(torch.bfloat16 if args.dtype == "bf16" else torch.float32)
    margs = make_args(hific_args, batch_size=1, image_dims=(3, S, S), latent_dims=(220, S // 16, S // 16))
    torch.manual_seed(0)
    model = hific_amd.Model(margs, model_type=ModelTypes.COMPRESSION_GAN, model_mode=ModelModes.EVALUATION,
                            allow_random_lpips_backbone=True, build_tables=True).to(dev).eval()
    x = torch.rand((1, 3, S, S), generator=torch.Generator(device=dev).manual_seed(11), device=dev)
    host = {"t": 0.0}

    def timed_fn(fn):
        def wrap(*a, **k):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = fn(*a, **k)
            host["t"] += time.perf_counter() - t0
            return r
        return wrap
    orig = (rans.ans_compress, rans.ans_decompress)
    rans.ans_compress, rans.ans_decompress = timed_fn(orig[0]), timed_fn(orig[1])
    codec.rans = rans
    try:
        res = {}
        out = model.compress(x, silent=True)                      # warm-up (packs, tables on the device)
        model.decompress(out)
        n = 3
        for name, fn in (("compress", lambda: model.compress(x, silent=True)), ("decompress", lambda: model.decompress(out))):
            host["t"] = 0.0
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(n):
                r = fn()
            torch.cuda.synchronize()
            tot = (time.perf_counter() - t0) / n
            res[f"{name}_ms"] = round(tot * 1e3, 2)
            res[f"{name}_host_rans_ms"] = round(host["t"] / n * 1e3, 2)
            res[f"{name}_device_ms"] = round((tot - host["t"] / n) * 1e3, 2)
        nbytes = 4 * (len(out.hyperlatents_encoded) + len(out.latents_encoded))
        res.update(workload=f"Model.compress / Model.decompress, one {S}x{S} RGB image ({S * S / 1e6:.2f} MP), {args.dtype} "
                            f"device half, vectorised rANS on the host (random-init weights: the bitstream is incompressible "
                            f"noise, {nbytes / 1e3:.0f} kB)",
                   bpp=round(8.0 * nbytes / (S * S), 3),
                   reference_note="the reference reports 2-3 s to decode a ~megapixel image on a GPU without the rANS coder "
                                  "(src/README.md:87)")
        return res
    finally:
        rans.ans_compress, rans.ans_decompress = orig


# ---- parity of the benchmarked mode, measured in this process ---------------------------------------------------------
def parity_leg(args, dev):
    """The benchmarked mode against the oracle (the checker, CPU float32) on the benchmark's own shape: same seeded
    weights, images and quantisation noise through hific_amd.Model (G-turn forward of the headline model) and through
    oracle.model_forward.  north_star: quantised latent indices bit-exact, outputs within 1e-3."""
    import numpy as np
    import torch
    import hific_amd
    from hific_amd import ops as hops
    from hific_amd.default_config import make_args, mse_lpips_args, hific_args, ModelTypes
    from oracle import hific_oracle as O
    gan = args.config == "gan"
    B, S = args.batch, args.size
    hific_amd.set_compute_dtype(torch.bfloat16 if args.dtype == "bf16" else torch.float32)
    margs = make_args(hific_args if gan else mse_lpips_args, regime=args.regime, batch_size=B, image_dims=(3, S, S),
                      latent_dims=(220, S // 16, S // 16))
    model = hific_amd.Model(margs, model_type=ModelTypes.COMPRESSION_GAN if gan else ModelTypes.COMPRESSION,
                            allow_random_lpips_backbone=True)
    sd, bb = O.make_state_dict(seed=0, gan=gan), O.make_alex_backbone()
    model.load_state_dict(sd, strict=True)
    model.perceptual_loss.load_backbone_state_dict(bb)
    model = model.to(dev).train()
    model.Hyperprior.keep_debug = True
    x = O.make_image(3, B, S, S)
    nh, nl = O.make_noise(6, (B, 320, S // 64, S // 64)), O.make_noise(7, (B, 220, S // 16, S // 16))
    noises = [nh.to(dev), nl.to(dev)]
    model.Hyperprior._draw_noise = lambda t: noises.pop(0)
    with torch.no_grad():
        losses, inter = model(x.to(dev), train_generator=True, return_intermediates=True, writeout=False)
    torch.cuda.synchronize()
    sym = torch.round(inter.latents_quantized.float() - model.Hyperprior.debug_latent_means).cpu()
    got = dict(loss=float(losses["compression"]), n_bpp=float(inter.n_bpp), q_bpp=float(inter.q_bpp))
    rec = inter.reconstruction.float().cpu()
    dec = inter.latents_quantized.detach().float()
    rec_x = None
    if args.dtype == "bf16":
        # the exact-reconstruction option (ops.set_exact_reconstruction): the Generator on the same decoded latents, no-grad,
        # split-bf16 contractions on float32 activations - what Model.decompress / the EVALUATION forward run under the option
        hops.set_exact_reconstruction(True)
        try:
            with torch.no_grad():
                rec_x = model.Generator(dec.contiguous()).float().cpu()
        finally:
            hops.set_exact_reconstruction(False)
    rec_t = None
    if args.dtype == "bf16":
        # the exact-TRAINING option (ops.set_exact_training): the same forward with autograd enabled - the mode `exact_training`
        # below times; its reconstruction comes from this training forward (same noise), indices compared as above
        hops.set_exact_training(True)
        try:
            noises = [nh.to(dev), nl.to(dev)]
            losses_t, inter_t = model(x.to(dev), train_generator=True, return_intermediates=True, writeout=False)
            same_idx = bool(torch.equal(inter_t.latents_quantized.detach().float(), dec))
            rec_t = inter_t.reconstruction.detach().float().cpu() if same_idx else None
            got_t = dict(loss=float(losses_t["compression"].detach()), n_bpp=float(inter_t.n_bpp), q_bpp=float(inter_t.q_bpp),
                         same_indices_as_default_mode=same_idx)
            del inter_t, losses_t
        finally:
            hops.set_exact_training(False)
    dec = dec.cpu()
    del model, losses, inter
    hops.pack_cache.clear(); hops.split_weights.clear()
    torch.cuda.empty_cache()
    lw = np.load(os.path.join(ROOT, "high-fidelity-generative-compression_amd", "loss", "weights", "lpips_alex_lin_v0.1.npz"))
    lins = [torch.from_numpy(lw[f"lin{i}"].copy()) for i in range(5)]
    oargs = dict(lambda_A=margs.lambda_A, target_rate=margs.target_rate)
    with torch.no_grad():
        out = O.model_forward(sd, bb, lins, x, step_counter=1, training=True, gan=gan, train_generator=True,
                              noise_hyper=nh, noise_latent=nl, args=oargs)
    hi = out["hyperinfo"]
    sym_o = torch.floor(out["y"] - hi.latent_means + 0.5)
    flips = sym != sym_o
    nflip = int(flips.sum())
    frac = out["y"] - hi.latent_means + 0.5
    frac = frac - torch.floor(frac)
    tie = torch.minimum(frac, 1 - frac)
    rel = lambda a, b: abs(a - b) / max(abs(b), 1e-30)
    res = {"against": "oracle (CPU float32 restatement of the reference), same weights / images / noise, "
                      f"batch {B} x {S}x{S}, {'compression_gan' if gan else 'compression'} G-turn forward",
           "mode": f"{args.dtype}" + (" + exact-index chain" if args.dtype == "bf16" and hops.exact_index_on() else ""),
           "n_indices": flips.numel(), "index_flips": nflip, "index_flip_rate": nflip / flips.numel(),
           "max_tie_distance_of_flips": float(tie[flips].max()) if nflip else 0.0,
           "flips_by_more_than_one": int(((sym - sym_o).abs() > 1).sum()),
           "loss_rel": rel(got["loss"], float(out["compression"])), "nbpp_rel": rel(got["n_bpp"], float(hi.total_nbpp)),
           "qbpp_rel": rel(got["q_bpp"], float(hi.total_qbpp))}
    # reconstruction GIVEN EQUAL INDICES (one flipped index moves a decoded latent by 1.0): the oracle Generator on the device's
    # decoded latents when a rounding tie went the other way
    ref = out["reconstruction"]
    if nflip:
        with torch.no_grad():
            ref = O.generator_forward(sd, dec, margs.n_residual_blocks)
    res["recon_rel"] = float((rec - ref).abs().max() / ref.abs().max())
    res["recon_rms_rel"] = float((rec - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    if rec_x is not None:
        res["recon_rel_exact_reconstruction_option"] = float((rec_x - ref).abs().max() / ref.abs().max())
        res["recon_note"] = ("recon_rel: max |rec - rec_oracle| / max |rec_oracle| of the benchmarked mode (bf16 Generator "
                             "activations); *_exact_reconstruction_option: the same latents through the Generator under "
                             "hific_amd.set_exact_reconstruction(True) (no-grad forwards only: decompress / EVALUATION), the "
                             "mode that meets north_star's 1e-3 on the reconstruction; its cost is fwd.exact_reconstruction_*")
    rnd = lambda d: {k: (float(f"{v:.4g}") if isinstance(v, float) else v) for k, v in d.items()}
    if rec_t is not None:
        res["recon_rel_exact_training_option"] = float((rec_t - ref).abs().max() / ref.abs().max())
        # the full-parity bf16 TRAINING mode (hific_amd.set_exact_training(True)) with its own block: same indices as the
        # default mode by construction (the Encoder / hyper chain is the same), every float output within north_star's 1e-3
        res["exact_training"] = rnd({
            "mode": "bf16 + exact-index chain + exact Generator chain (hific_amd.set_exact_training(True)): float32-accurate "
                    "forward values, bf16 stored activations, bf16 backward",
            "index_flips": nflip, "same_indices_as_default_mode": got_t["same_indices_as_default_mode"],
            "loss_rel": rel(got_t["loss"], float(out["compression"])), "nbpp_rel": rel(got_t["n_bpp"], float(hi.total_nbpp)),
            "qbpp_rel": rel(got_t["q_bpp"], float(hi.total_qbpp)),
            "recon_rel": float((rec_t - ref).abs().max() / ref.abs().max()),
            "recon_rms_rel": float((rec_t - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()),
            "gradients": "every G-turn parameter gradient vs the float32 oracle under fixed per-class ceilings (2e-2 ... 3e-2): "
                         "tests/test_gpu_fullsize_backward.py::test_bf16_modes_every_G_turn_gradient_against_the_oracle_fullsize",
            "meets_north_star_1e-3": bool(float((rec_t - ref).abs().max() / ref.abs().max()) < 1e-3
                                          and rel(got_t["loss"], float(out["compression"])) < 1e-3)})
    res["product_default"] = ("bf16 + exact-index chain with the plain bf16 Generator (the headline `value`): indices exact up to "
                              "rounding ties, rates / loss within 1e-3, reconstruction at bf16-activation distance (recon_rel); "
                              "the mode that meets the whole parity clause is `exact_training` (own throughput leg, own block here)")
    return rnd(res)


# ---------------------------------------------------------------------------------------------------------------
def respawn(args):
    """`python bench.py --gpus N` from a plain shell: become N ranks under torch.distributed.run."""
    port = 29500 + (os.getpid() % 400)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    sys.exit(subprocess.call(cmd, env=env))


def main():
    args = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and env_world is None:
        respawn(args)
    import torch
    import torch.distributed as dist
    world = int(env_world or "1")
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or os.environ.get("HIFIC_FORCE_DIST") == "1"   # the latter: 1-rank RCCL smoke test
    # The contract is ONE JSON line on stdout.  Native libraries write there too (RCCL prints a five-line version banner to
    # fd 1 when its first communicator comes up): everything but the line itself goes to stderr.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(obj), flush=True)
        os.dup2(2, 1)

    # HIFIC_BENCH_REHEARSAL=1: the multi-rank control flow on a ONE-GPU box - every rank on cuda:0, gloo as the transport (RCCL
    # refuses two ranks on one device).  Same reducers, same collectives in the same order, same report; the numbers mean nothing.
    rehearsal = os.environ.get("HIFIC_BENCH_REHEARSAL") == "1" and world > 1
    if rehearsal:
        local = 0
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # (read when the runtime comes up: before the first device call)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    cfg = args.config
    model, opts, reducers = build(args, dev, cfg)
    step = make_step(args, model, opts, reducers, dev, cfg)
    if args.traffic_child:                       # counter pass of measure_traffic(): a few steps, nothing else
        timed(step, args.steps, args.warmup, fence)
        return
    if use_dist:
        for r in reducers.values():
            r.measure_exposed(True)              # two event records per finish(): read after the timed region
    for _ in range(args.warmup):                 # (choose_launch needs a settled model: packs, arenas, workspaces)
        step()
    run_step, is_graph, launch_cal = choose_launch(args, step, world, fence)
    elapsed = timed(run_step, args.steps, args.warmup, fence)
    rccl = None
    if use_dist:
        # the job the driver launched is the job the collectives run in
        assert dist.get_backend() == ("gloo" if rehearsal else "nccl") and \
            dist.get_world_size() == world == int(os.environ.get("WORLD_SIZE", "1")), \
            (dist.get_backend(), dist.get_world_size(), world, os.environ.get("WORLD_SIZE"))
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # (warm-up steps are included in the event pairs: average over all measured steps)
        exp_ms = sum(r.exposed_comm_ms() for r in reducers.values()) / (args.steps + args.warmup)
        t = torch.tensor([exp_ms], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        rccl = {"backend": "gloo (REHEARSAL on one device: control flow only)" if rehearsal else "nccl (RCCL)", "rccl_ranks": world,
                "buckets": {k: len(r.buckets) for k, r in reducers.items()},
                "bucket_mbytes": float(os.environ.get("HIFIC_BUCKET_MB", 128)),
                "bucket_tail_mbytes": os.environ.get("HIFIC_BUCKET_TAIL_MB", "2,32"),
                "gradient_payload": next(iter(reducers.values())).payload,
                "gradient_mbytes_per_step": {k: round(r.arena.numel * 4 / 2 ** 20, 1) for k, r in reducers.items()},
                "exposed_comm_ms": round(float(t.item()), 3),
                "exposed_comm_note": "GPU time per step the compute stream waits in BucketedGradReducer.finish() for "
                                     "collectives the backward pass did not hide (max over ranks)"}
        rccl.update(scale_report(args, step, reducers, opts, world, rank, dev, fence, elapsed / args.steps * 1e3))
    imgs_per_step = args.batch * (2 if cfg == "gan" else 1)
    value = world * imgs_per_step * args.steps / elapsed
    ms_step = elapsed / args.steps * 1e3
    tflop_step = (2 * (GMAC_G_TURN + GMAC_D_TURN) if cfg == "gan" else GFLOP_PER_IMAGE["compression"]) \
        * args.batch * (args.size / 256.0) ** 2 / 1e3
    peak = MFMA_PEAK_TFLOPS[args.dtype]
    what = ("compression_gan G-turn + D-turn cycle (2 batches)" if cfg == "gan" else "compression training step")
    out = {
        "metric": "training images/sec (256x256) HiFIC-low", "value": round(value, 3), "unit": "images/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"HiFIC-{args.regime} {what}, batch {args.batch}/GPU/turn, {args.size}x{args.size} RGB, "
                               f"{args.dtype} MFMA compute, f32 master weights, fwd+bwd+Adam, random-init weights "
                               f"(BASELINE configs[{2 if cfg == 'gan' else 1}])",
                   "images_per_step": world * imgs_per_step, "global_batch": world * args.batch,
                   "parallelism": f"dp{world}"},
        "step_model": {"algorithmic_tflop_per_step_per_gpu": round(tflop_step, 3),
                       "whole_step_tflops_per_gpu": round(tflop_step / (ms_step * 1e-3), 1),
                       "whole_step_frac_of_mfma_peak": round(tflop_step / (ms_step * 1e-3) / peak, 4)},
    }
    if args.dtype == "bf16":
        from hific_amd import ops as _ops
        out["config"]["exact_index_chain"] = bool(_ops.exact_index_on())
    out["config"]["launch"] = ("one single-stream hipGraph replay per cycle" if is_graph else
                               "eager (one launch per kernel, side / branch streams)")
    if launch_cal is not None:
        out["launch_modes"] = launch_cal
    if rccl is not None:
        out["rccl"] = rccl
    extras = world == 1 and not args.no_extras
    if extras:
        from hific_amd import ops as hific_ops
        # ---- roofline of the dominant kernel function of the headline step (live HIP events) --------------------
        prof = profile_kernels(step, max(1, min(args.steps, 4)))
        dom = max(prof, key=lambda k: prof[k]["ms_per_step"])
        d = prof[dom]
        traffic, why = (None, "skipped")
        out["roofline"] = {
            "bound": "mfma", "kernel": dom, "achieved": round(d["tflops"], 2), "peak": peak, "unit": "TFLOP/s",
            "frac": round(d["tflops"] / peak, 4), "avg_launch_us": round(d["avg_launch_us"], 2),
            "algorithmic_gflop_per_launch": round(d["gflop_per_launch"], 3),
            "launches_per_step": d["launches_per_step"], "traffic": None,
            "per_kernel": {k: {kk: ({a: round(b, 3) for a, b in vv.items()} if isinstance(vv, dict) else round(vv, 3))
                                   for kk, vv in v.items()} for k, v in
                           sorted(prof.items(), key=lambda kv: -kv[1]["ms_per_step"])},
            "gemm_class_ms_per_step": round(sum(v["ms_per_step"] for v in prof.values()), 3),
        }
        if os.environ.get("HIFIC_BENCH_ROOFLINE_ONLY") == "1":      # kernel experiments: headline + per-kernel table, nothing else
            emit(out)
            return
        del model, opts, reducers, step, run_step
        hific_ops.pack_cache.clear(); hific_ops.split_weights.clear()
        torch.cuda.empty_cache()
        # ---- configs[1]: compression (no GAN) training step ---------------------------------------------------
        if cfg == "gan":
            m2, o2, r2 = build(args, dev, "compression")
            s2 = make_step(args, m2, o2, r2, dev, "compression")
            for _ in range(args.warmup):
                s2()
            s2, g2, _ = choose_launch(args, s2, world, fence)
            e2 = timed(s2, args.steps, args.warmup, fence)
            out["compression"] = {"value": round(args.batch * args.steps / e2, 3), "unit": "images/s",
                                  "launch": "graph" if g2 else "eager",
                                  "ms_per_step": round(e2 / args.steps * 1e3, 3),
                                  "workload": "BASELINE configs[1]: compression (no GAN) training step, same batch/dtype",
                                  "whole_step_tflops": round(GFLOP_PER_IMAGE["compression"] * args.batch *
                                                             (args.size / 256.0) ** 2 / 1e3 / (e2 / args.steps), 1)}
            del m2, o2, r2, s2
            hific_ops.pack_cache.clear(); hific_ops.split_weights.clear()
            torch.cuda.empty_cache()
        # ---- forward ms/image: EVALUATION-mode forward (model.py:357-366) ---------------------------------------
        import hific_amd
        from hific_amd.default_config import make_args, hific_args, ModelTypes, ModelModes
        margs = make_args(hific_args, batch_size=args.batch, image_dims=(3, args.size, args.size),
                          latent_dims=(220, args.size // 16, args.size // 16))
        torch.manual_seed(0)
        ev = hific_amd.Model(margs, model_type=ModelTypes.COMPRESSION_GAN, model_mode=ModelModes.EVALUATION,
                             allow_random_lpips_backbone=True, build_tables=False).to(dev).eval()
        xg = torch.Generator(device=dev).manual_seed(7)
        xe = torch.rand((args.batch, 3, args.size, args.size), generator=xg, device=dev)

        def fwd():
            with torch.no_grad():
                return ev(xe)
        fwd.generators = ()
        for _ in range(3):
            fwd()
        fwd_run, gf, _ = choose_launch(args, fwd, world, fence)
        ef = timed(fwd_run, max(args.steps, 10), args.warmup, fence)
        per_img = ef / max(args.steps, 10) / args.batch * 1e3
        out["fwd_ms_per_image"] = round(per_img, 4)
        out["fwd"] = {"workload": f"EVALUATION-mode Model.forward (Encoder -> Hyperprior -> Generator, clamp; returns "
                                  f"reconstruction + q_bpp), no-grad, batch {args.batch}, {args.dtype}",
                      "launch": "graph" if gf else "eager",
                      "ms_per_batch": round(per_img * args.batch, 3), "images_per_s": round(1e3 / per_img, 1),
                      "tflops": round(GFLOP_PER_IMAGE["forward"] * (args.size / 256.0) ** 2 / 1e3 / (per_img * 1e-3), 1)}
        if args.dtype == "bf16" and hific_ops.exact_index_on():
            hific_ops.set_exact_index(False)                     # for comparison: the plain bf16 chain (0.39 % index flips)
            try:
                fwd(); fwd()
                fwd2, _, _ = choose_launch(args, fwd, world, fence)
                ef2 = timed(fwd2, max(args.steps, 10), 2, fence)
                del fwd2
            finally:
                hific_ops.set_exact_index(True)
            out["fwd"]["plain_bf16_chain_ms_per_image"] = round(ef2 / max(args.steps, 10) / args.batch * 1e3, 4)
            hific_ops.set_exact_reconstruction(True)             # the option that reconstructs within 1e-3 (parity.recon_rel_*)
            try:
                fwd(); fwd()
                fwd3, _, _ = choose_launch(args, fwd, world, fence)
                ef3 = timed(fwd3, max(args.steps, 10), 2, fence)
                del fwd3
            finally:
                hific_ops.set_exact_reconstruction(False)
            out["fwd"]["exact_reconstruction_ms_per_image"] = round(ef3 / max(args.steps, 10) / args.batch * 1e3, 4)
        del ev
        hific_ops.pack_cache.clear(); hific_ops.split_weights.clear()
        torch.cuda.empty_cache()
        default_shape = args.size == 256 and args.regime == "low"
        # ---- the same cycle under the exact-TRAINING option: split-bf16 Generator forward with autograd (float32 Generator
        #      activations, bf16 MFMA everywhere): what north_star's 1e-3 on the reconstruction costs a bf16 training step ----
        if args.dtype == "bf16" and default_shape:
            hific_ops.set_exact_training(True)
            try:
                m4, o4, r4 = build(args, dev, cfg)
                s4 = make_step(args, m4, o4, r4, dev, cfg)
                for _ in range(3):                   # (the first cycles create this mode's pack-cache entries)
                    s4()
                n4 = max(4, min(args.steps, 8))
                e4 = timed(s4, n4, 1, fence)
                out["exact_training"] = {
                    "value": round(imgs_per_step * n4 / e4, 3), "unit": "images/s", "ms_per_step": round(e4 / n4 * 1e3, 3),
                    "launch": "eager",
                    "workload": "the headline cycle with hific_amd.set_exact_training(True) - the full-parity bf16 mode: "
                                "exact-index chain + the Generator as an exact chain (split-bf16 contractions, float32-accurate "
                                "forward values, nominal bf16 activations, the plain bf16 backward); its parity block: "
                                "parity.exact_training; gradients vs the float32 oracle under fixed ceilings: "
                                "tests/test_gpu_fullsize_backward.py::test_bf16_modes_every_G_turn_gradient_against_the_oracle_fullsize"}
                del m4, o4, r4, s4
            finally:
                hific_ops.set_exact_training(False)
            hific_ops.pack_cache.clear(); hific_ops.split_weights.clear()
            torch.cuda.empty_cache()
        # ---- the same cycle in float32 parity mode (f32 MFMA: exact fma chains) ---------------------------------------
        if args.dtype == "bf16" and default_shape:
            a32 = argparse.Namespace(**vars(args)); a32.dtype = "f32"
            m3, o3, r3 = build(a32, dev, cfg)
            s3 = make_step(a32, m3, o3, r3, dev, cfg)
            s3(); s3()
            s3, g3, _ = choose_launch(a32, s3, world, fence, ncal=2)
            n3 = max(2, min(args.steps, 4))
            e3 = timed(s3, n3, 1, fence)
            out["f32"] = {"value": round(imgs_per_step * n3 / e3, 3), "unit": "images/s", "ms_per_step": round(e3 / n3 * 1e3, 3),
                          "launch": "graph" if g3 else "eager",
                          "workload": "the headline cycle in float32 parity mode (v_mfma_f32_32x32x2_f32, f32 activations): "
                                      "every output within 1e-3 of the oracle (tests/test_gpu_golden.py, "
                                      "tests/test_gpu_fullsize_backward.py)",
                          "whole_step_frac_of_f32_mfma_peak": round(tflop_step / (e3 / n3) / MFMA_PEAK_TFLOPS["f32"], 4)}
            del m3, o3, r3, s3
            hific_amd.set_compute_dtype(torch.bfloat16)
            hific_ops.pack_cache.clear(); hific_ops.split_weights.clear()
            torch.cuda.empty_cache()
        # ---- BASELINE configs[4] per-GPU shape: regime high, one 1024 x 1024 crop per turn ---------------------------
        if cfg == "gan" and default_shape:
            a5 = argparse.Namespace(**vars(args)); a5.size, a5.batch, a5.regime = 1024, 1, "high"
            m5, o5, r5 = build(a5, dev, "gan")
            s5 = make_step(a5, m5, o5, r5, dev, "gan")
            s5(); s5()
            s5, g5, _ = choose_launch(a5, s5, world, fence)
            n5 = max(3, min(args.steps, 10))
            e5 = timed(s5, n5, 2, fence)
            out["config5_1gpu"] = {"value": round(2 * n5 / e5, 3), "unit": "images/s (1024x1024)",
                                   "ms_per_step": round(e5 / n5 * 1e3, 3), "launch": "graph" if g5 else "eager",
                                   "workload": f"BASELINE configs[4] on one GPU: compression_gan regime high, G-turn + D-turn "
                                               f"cycle, 1 x 1024x1024 crop per turn, {args.dtype}, fwd+bwd+Adam",
                                   "megapixels_per_s": round(2 * n5 * 1.048576 / e5, 2),
                                   "whole_step_tflops": round(tflop_step / args.batch * 16 / (e5 / n5), 1)}
            del m5, o5, r5, s5
            hific_ops.pack_cache.clear(); hific_ops.split_weights.clear()
            torch.cuda.empty_cache()
        # ---- the EVALUATION path around the hot path: compress / decompress of one megapixel image ---------------------
        if default_shape and os.environ.get("HIFIC_BENCH_NO_CODEC") != "1":
            try:
                out["codec_1mp"] = codec_leg(args, dev)
            except Exception as e:
                out["codec_1mp"] = {"error": f"{type(e).__name__}: {e}"}
            hific_ops.pack_cache.clear(); hific_ops.split_weights.clear()
            torch.cuda.empty_cache()
        # ---- parity of the benchmarked mode vs the oracle, at the benchmarked shape ---------------------------------
        if not args.no_parity:
            try:
                out["parity"] = parity_leg(args, dev)
            except Exception as e:                      # the throughput line must survive a checker failure
                out["parity"] = {"error": f"{type(e).__name__}: {e}"}
        # the small-layer launches as one class: hyperprior nets, LPIPS 15 x 15 layers, the
        # Discriminator's context / output convs, the 960 -> 220 layer and their gradients - grids of 32-400 workgroups bound by
        # neither roofline
        small_keys = [k for k in prof if k in ("gconv_kernel<bf16,64,1,4,1,1>", "gconv_kernel<bf16,64,2,2,1,2>",
                                                "gconv_kernel<bf16,64,2,2,2,2>", "gconv_sp9_kernel<2,4> narrow",
                                                "gconv_sp9_kernel<1,1> narrow", "gconv_sp9_kernel<2,4,rfx> narrow",
                                                "wgrad_kernel<bf16>")]
        if small_keys:
            ms_s = sum(prof[k]["ms_per_step"] for k in small_keys)
            n_s = sum(prof[k]["launches_per_step"] for k in small_keys)
            gf_s = sum(prof[k]["gflop_per_launch"] * prof[k]["launches_per_step"] for k in small_keys)
            out["roofline"]["small_layers"] = {
                "kernels": small_keys, "launches_per_step": round(n_s, 1), "ms_per_step": round(ms_s, 3),
                "tflops": round(gf_s / ms_s, 1) if ms_s > 0 else None,
                "frac_of_mfma_peak": round(gf_s / ms_s / peak, 4) if ms_s > 0 else None}
        pp = practical_peak()
        if pp is not None:
            out["roofline"]["practical_peak"] = pp
            out["roofline"]["frac_of_practical_peak"] = round(out["roofline"]["achieved"] / pp["tflops"], 4)
        if not args.no_traffic and os.environ.get("HIFIC_BENCH_PMC", "1") != "0":
            traffic, why = measure_traffic(args, dom)
        out["roofline"]["traffic"] = traffic
        if traffic is None:
            out["roofline"]["traffic_note"] = why
        else:
            us = out["roofline"]["avg_launch_us"]
            out["roofline"]["hbm_gbps_at_avg_launch"] = round(traffic["bytes_per_launch"] / (us * 1e-6) / 1e9, 1)
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline and not args.no_extras:
            out["cpu_baseline"] = cpu_baseline(args)
        emit(out)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

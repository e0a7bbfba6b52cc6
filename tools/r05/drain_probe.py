"""Is the GPU ever waiting for the host inside the headline cycle?  At marked points of the eager cycle the main stream is queried
(non-blocking): True = everything enqueued so far has finished = the GPU is idle until the host issues the next launch.  Also
host wall-clock per segment.  usage: python tools/r05/drain_probe.py"""
import os, sys, time, collections
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
import bench
sys.argv = [sys.argv[0]]
args = bench.parse()
dev = torch.device("cuda:0")
model, opts, reducers = bench.build(args, dev, "gan")
gen = torch.Generator(device=dev).manual_seed(1234)
B, S = args.batch, args.size
drained = collections.Counter(); seg = collections.defaultdict(float); last = [0.0]
def mark(name):
    t = time.perf_counter()
    seg[name] += t - last[0]; last[0] = t
    if torch.cuda.current_stream().query(): drained[name] += 1
def batch(): return torch.rand((B, 3, S, S), generator=gen, device=dev, dtype=torch.float32)
def rs(names):
    for n in names:
        opts[n].grad_scale = reducers[n].finish(); opts[n].step(); opts[n].zero_grad()
def step(probe):
    m = mark if probe else (lambda n: None)
    last[0] = time.perf_counter()
    x = batch(); m("G batch")
    losses = model(x, train_generator=True, writeout=False); m("G forward")
    losses["compression"].backward(); m("G backward")
    rs(["amort", "hyper"]); m("G optimizer")
    x = batch(); m("D batch")
    losses = model(x, train_generator=False, writeout=False); m("D forward")
    losses["disc"].backward(); m("D backward")
    rs(["disc"]); opts["amort"].zero_grad(); opts["hyper"].zero_grad(); m("D optimizer")
for _ in range(5): step(False)
torch.cuda.synchronize()
N = 20
t0 = time.perf_counter()
for _ in range(N): step(True)
th = time.perf_counter() - t0
torch.cuda.synchronize(); te = time.perf_counter() - t0
print(f"host {th / N * 1e3:.2f} ms/cycle, end to end {te / N * 1e3:.2f} ms/cycle")
for k in ("G batch", "G forward", "G backward", "G optimizer", "D batch", "D forward", "D backward", "D optimizer"):
    print(f"  after {k:12s}: host {seg[k] / N * 1e3:6.2f} ms, main stream drained in {drained[k]:2d} of {N} cycles")

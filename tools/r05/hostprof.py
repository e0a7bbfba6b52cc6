"""cProfile of the host side of the headline cycle (eager, multi-stream): where the ~17 ms of enqueueing per cycle go."""
import os, sys, cProfile, pstats, io, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
import bench
sys.argv = [sys.argv[0]]
args = bench.parse()
dev = torch.device("cuda:0")
model, opts, reducers = bench.build(args, dev, "gan")
step = bench.make_step(args, model, opts, reducers, dev, "gan")
for _ in range(4): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): step()
th = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"host enqueue {th / 5 * 1e3:.2f} ms/cycle, total {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms/cycle")
pr = cProfile.Profile()
pr.enable()
for _ in range(5): step()
pr.disable()
torch.cuda.synchronize()
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
    print(s.getvalue()[:9000])

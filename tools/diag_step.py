"""Diagnostic: time the phases of one training step with syncs (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

class A: pass
args = A(); args.batch = int(os.environ.get("B", 16)); args.size = 256; args.config = os.environ.get("CFG", "compression"); args.dtype = os.environ.get("DT", "bf16")
dev = torch.device("cuda:0")
def T(msg, t0):
    torch.cuda.synchronize(); print(f"{msg}: {time.time()-t0:.3f}s", flush=True); return time.time()
t = time.time()
model, opts, reducers = bench.build(args, dev)
t = T("build", t)
x = torch.rand((args.batch, 3, 256, 256), device=dev)
for it in range(3):
    t = time.time()
    y = model.Encoder(x); t = T(f"[{it}] encoder fwd", t)
    hi = model.Hyperprior(y, spatial_shape=(256, 256)); t = T(f"[{it}] hyperprior fwd", t)
    xg = model.Generator(hi.decoded); t = T(f"[{it}] generator fwd", t)
    mse = model.distortion_loss(xg, x); t = T(f"[{it}] mse", t)
    lp = model.perceptual_loss_wrapper(xg, x); t = T(f"[{it}] lpips fwd", t)
    loss = hi.total_nbpp + 0.002 * mse + lp
    loss.backward(); t = T(f"[{it}] backward", t)
    for o in opts.values():
        o.step(); o.zero_grad()
    t = T(f"[{it}] adam", t)
print("mem GB", torch.cuda.max_memory_allocated() / 2**30)

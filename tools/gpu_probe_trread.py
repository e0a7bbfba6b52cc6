"""Hardware-semantics probe (not a pytest): checks the ds_read_b64_tr_b16 lane mapping assumed by the bf16
weight-gradient kernel and the MFMA fragment layouts, by running tiny wgrad/conv cases whose answers are known."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hific_amd
from hific_amd import ops, lib
import torch.nn.functional as F

dev = torch.device("cuda:0")
for dt in (torch.float32, torch.bfloat16):
    hific_amd.set_compute_dtype(dt)
    g = torch.Generator().manual_seed(0)
    x = (torch.rand((1, 40, 8, 8), generator=g) * 2 - 1).to(dt).float()
    w = (torch.rand((70, 40, 3, 3), generator=g) * 2 - 1).to(dt).float()
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, padding=1)
    gy = (torch.rand(yr.shape, generator=g) * 2 - 1).to(dt).float()
    yr.backward(gy)
    xd, wd = x.to(dev).to(dt).requires_grad_(True), w.to(dev).requires_grad_(True)
    y = ops.conv2d(xd, wd, None, 1, (1, 1, 1, 1), lib.PAD_ZERO)
    y.backward(gy.to(dev).to(dt))
    torch.cuda.synchronize()
    def rel(a, b): return ((a - b).abs().max() / b.abs().max()).item()
    print(dt, "fwd", rel(y.detach().float().cpu(), yr.detach()), "dx", rel(xd.grad.float().cpu(), xr.grad),
          "dw", rel(wd.grad.cpu(), wr.grad))
    if rel(wd.grad.cpu(), wr.grad) > 0.05:
        d = wd.grad.cpu()
        print("  dw mismatch sample got", d[0, 0].flatten()[:5].tolist(), "want", wr.grad[0, 0].flatten()[:5].tolist())

"""Micro-benchmark of single conv ops through the C-ABI (for rocprofv3 counter passes and A/B timing).
usage: python tools/micro_conv.py [fwd|bwd|wgrad|all] [iters]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hific_amd
from hific_amd import ops, lib

which = sys.argv[1] if len(sys.argv) > 1 else "all"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
hific_amd.set_compute_dtype(torch.bfloat16)
N, C, K, H = 16, 960, 960, 16
x = torch.randn(N, C, H, H, device=dev).bfloat16().requires_grad_(True)
w = (torch.randn(K, C, 3, 3, device=dev) * 0.01).requires_grad_(True)
b = torch.zeros(K, device=dev, requires_grad=True)
gy = torch.randn(N, K, H, H, device=dev).bfloat16()
ws = lib.workspace(dev)
def run(name, fn):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / iters
    fl = 2.0 * N * H * H * K * C * 9
    print(f"{name}: {dt*1e6:.1f} us  ({fl/dt/1e12:.1f} TFLOP/s incl. pack)", flush=True)
geom = (N, C, H, H, K, 3, 3, 1, 1, 1, 1, 1, lib.PAD_REFLECT)
y = torch.empty(N, K, H, H, device=dev, dtype=torch.bfloat16)
dx = torch.empty_like(x)
dw = torch.empty_like(w)
if which in ("fwd", "all"):
    run("fwd", lambda: lib.call("hific_conv2d_fwd", x.data_ptr(), w.data_ptr(), None, b.data_ptr(), None, y.data_ptr(), *geom, 0, 1, 0, ws.data_ptr(), ws.numel(), lib.stream()))
if which in ("bwd", "all"):
    run("bwd_data", lambda: lib.call("hific_conv2d_bwd_data", gy.data_ptr(), w.data_ptr(), None, dx.data_ptr(), *geom, 1, 0, ws.data_ptr(), ws.numel(), lib.stream()))
if which in ("wgrad", "all"):
    run("bwd_weight", lambda: lib.call("hific_conv2d_bwd_weight", x.data_ptr(), gy.data_ptr(), dw.data_ptr(), *geom, 0, 1, 0, ws.data_ptr(), ws.numel(), lib.stream()))

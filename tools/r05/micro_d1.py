"""Timing of the pieces of the Discriminator input stage backward (tools/r05)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import hific_amd
from hific_amd import lib
dev = torch.device("cuda:0")
B, K, Ci, Cc, H, W, f = 16, 64, 3, 12, 256, 256, 16
dz = torch.randn(2 * B, K, H // 2, W // 2, device=dev).bfloat16()
w = torch.randn(K, Ci + Cc, 4, 4, device=dev) * 0.05
isg = torch.tensor([0.7], device=dev)
out = torch.empty(B, Cc, H // f, W // f, device=dev, dtype=torch.bfloat16)
ws = lib.workspace(dev)
def run(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
print("d1_ctx_grad: %.1f us" % run(lambda: lib.call("hific_d1_ctx_grad", dz.data_ptr(), w.data_ptr(), isg.data_ptr(), out.data_ptr(),
                                                     B, K, Ci, Cc, H, W, f, lib.dtype_code(dz), ws.data_ptr(), ws.numel(), lib.stream())))

"""Micro-benchmark of ChannelNorm fwd/bwd through the ops layer. env: MN MC MH, HIFIC_CN_{FWD,BWD}_SEL"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hific_amd
from hific_amd import ops
E = lambda k, d: int(os.environ.get(k, d))
N, C, H = E("MN", 16), E("MC", 960), E("MH", 16)
dev = torch.device("cuda:0")
x = torch.randn(N, C, H, H, device=dev).bfloat16().requires_grad_(True)
g = torch.ones(1, C, 1, 1, device=dev, requires_grad=True); b = torch.zeros(1, C, 1, 1, device=dev, requires_grad=True)
gy = torch.randn(N, C, H, H, device=dev).bfloat16()
def t(fn, it=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e6
with torch.no_grad():
    tf = t(lambda: ops.channel_norm(x, g, b, 1e-3, relu=True))
def fb():
    y = ops.channel_norm(x, g, b, 1e-3, relu=True); y.backward(gy)
tb = t(fb)
print(f"N{N} C{C} H{H} fsel={os.environ.get('HIFIC_CN_FWD_SEL')} bsel={os.environ.get('HIFIC_CN_BWD_SEL')}: fwd {tf:.1f} us, fwd+bwd {tb:.1f} us", flush=True)

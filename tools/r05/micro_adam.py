"""Adam kernel alone on the amortisation arena size (181.46 M parameters): us per launch and effective TB/s (28 B per parameter)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import hific_amd
from hific_amd import lib
dev = torch.device("cuda:0")
n = 181_460_000 // 4 * 4
p = torch.randn(n, device=dev); g = torch.randn(n, device=dev) * 1e-3; m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
bc = torch.tensor([0.1, 0.0316], device=dev)
def run():
    lib.call("hific_adam_apply", p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), n, 1e-4, 0.9, 0.999, 1e-8, bc.data_ptr(), 1.0, lib.stream())
for _ in range(3): run()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(20): run()
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
print(f"adam {dt*1e6:.1f} us  {n*28/dt/1e12:.2f} TB/s  checksum {float(p[:1000].sum()):.6f}")

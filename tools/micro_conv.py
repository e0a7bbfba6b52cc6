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
E = lambda k, d: int(os.environ.get(k, d))
N, C, K, H, R, ST = E("MN", 16), E("MC", 960), E("MK", 960), E("MH", 16), E("MR", 3), E("MS", 1)
if ST == 1: pads = (R // 2,) * 4
else: pads = (1, 0, 0, 1)                    # (top, left, bottom, right) of the encoder's asymmetric reflect pad
if os.environ.get("MPAD"): pads = tuple(int(v) for v in os.environ["MPAD"].split(","))
OH = (H + pads[0] + pads[2] - R) // ST + 1
x = torch.randn(N, C, H, H, device=dev).bfloat16().requires_grad_(True)
w = (torch.randn(K, C, R, R, device=dev) * 0.01).requires_grad_(True)
b = torch.zeros(K, device=dev, requires_grad=True)
gy = torch.randn(N, K, OH, OH, device=dev).bfloat16()
ws = lib.workspace(dev)
def run(name, fn):
    import ctypes
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / iters
    fl = 2.0 * N * OH * OH * K * C * R * R
    extra = ""
    if os.environ.get("MPROF") == "1":           # the in-library profiler: the GEMM kernel alone (HIP events around its launch)
        lib.call("hific_prof_begin")
        for _ in range(iters): fn()
        ms = (ctypes.c_double * 32)(); flp = (ctypes.c_double * 32)(); cnt = (ctypes.c_int * 32)()
        names = ctypes.create_string_buffer(32 * 64)
        nk = lib.raw("hific_prof_end")(32, ms, flp, cnt, names)
        for k in range(max(nk, 0)):
            if cnt[k]:
                extra += f"  [{names.raw[k * 64:(k + 1) * 64].split(bytes(1), 1)[0].decode()} {ms[k] * 1e3 / cnt[k]:.1f} us]"
    print(f"{name}: {dt*1e6:.1f} us  ({fl/dt/1e12:.1f} TFLOP/s incl. pack){extra}", flush=True)
geom = (N, C, H, H, K, R, R, ST, pads[0], pads[1], pads[2], pads[3], lib.PAD_REFLECT)
y = torch.empty(N, K, OH, OH, device=dev, dtype=torch.bfloat16)
dx = torch.empty_like(x)
dw = torch.empty_like(w)
if which in ("fwd", "all"):
    run("fwd", lambda: lib.call("hific_conv2d_fwd", x.data_ptr(), w.data_ptr(), None, b.data_ptr(), None, y.data_ptr(), *geom, 0, 1, 0, ws.data_ptr(), ws.numel(), None, 0, 0, lib.stream()))
if which in ("bwd", "all"):
    run("bwd_data", lambda: lib.call("hific_conv2d_bwd_data", gy.data_ptr(), w.data_ptr(), None, dx.data_ptr(), *geom, 1, 0, ws.data_ptr(), ws.numel(), None, 0, 0, lib.stream()))
if which == "convt":
    # nn.ConvTranspose2d k3 s2 p1 op1 (generator up-sampling): x [N,C,H,H] -> y [N,K,2H,2H]
    wt = (torch.randn(C, K, 3, 3, device=dev) * 0.01)
    yt = torch.empty(N, K, 2 * H, 2 * H, device=dev, dtype=torch.bfloat16)
    run("convt_fwd", lambda: lib.call("hific_conv_transpose2d_fwd", x.data_ptr(), wt.data_ptr(), b.data_ptr(), yt.data_ptr(),
                                      N, C, H, H, K, 3, 3, 2, 1, 1, 0, 1, 0, ws.data_ptr(), ws.numel(), None, 0, 0, lib.stream()))
if which in ("wgrad", "all"):
    run("bwd_weight", lambda: lib.call("hific_conv2d_bwd_weight", x.data_ptr(), gy.data_ptr(), dw.data_ptr(), *geom, 0, 1, 0, ws.data_ptr(), ws.numel(), lib.stream()))

"""Kernel-only timing (library HIP-event profiler) of the residual-block convolution, forward and reflect data gradient:
python tools/micro_sp9.py [iters]   env: MN MC MK MH as tools/micro_conv.py; kernel variants via HIFIC_SP9_* / HIFIC_LIB_PATH"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hific_amd
from hific_amd import lib

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")
hific_amd.set_compute_dtype(torch.bfloat16)
E = lambda k, d: int(os.environ.get(k, d))
N, C, K, H, R, ST = E("MN", 16), E("MC", 960), E("MK", 960), E("MH", 16), E("MR", 3), E("MS", 1)
pads = (R // 2,) * 4 if ST == 1 else (1, 0, 0, 1)      # stride 2: the Encoder's asymmetric reflect pad
OH = (H + pads[0] + pads[2] - R) // ST + 1
x = torch.randn(N, C, H, H, device=dev).bfloat16()
w = torch.randn(K, C, R, R, device=dev) * 0.01
b = torch.zeros(K, device=dev)
gy = torch.randn(N, K, OH, OH, device=dev).bfloat16()
y = torch.empty(N, K, OH, OH, device=dev, dtype=torch.bfloat16)
dx = torch.empty_like(x)
ws = lib.workspace(dev)
geom = (N, C, H, H, K, R, R, ST, pads[0], pads[1], pads[2], pads[3], lib.PAD_REFLECT)
only = os.environ.get("MOPS", "fwd,bwd").split(",")
fwd = lambda: lib.call("hific_conv2d_fwd", x.data_ptr(), w.data_ptr(), None, b.data_ptr(), None, y.data_ptr(), *geom, 0, 1, 0,
                       ws.data_ptr(), ws.numel(), None, 0, 0, lib.stream())
bwd = lambda: lib.call("hific_conv2d_bwd_data", gy.data_ptr(), w.data_ptr(), None, dx.data_ptr(), *geom, 1, 0, ws.data_ptr(),
                       ws.numel(), None, 0, 0, lib.stream())
_f, _b = fwd, bwd
fwd = _f if "fwd" in only else (lambda: None)
bwd = _b if "bwd" in only else (lambda: None)
for _ in range(3):
    fwd(); bwd()
torch.cuda.synchronize()
lib.call("hific_prof_begin")
for _ in range(iters):
    fwd(); bwd()
MAXK = 32
ms = (ctypes.c_double * MAXK)(); fl = (ctypes.c_double * MAXK)(); cnt = (ctypes.c_int * MAXK)()
names = ctypes.create_string_buffer(MAXK * 64)
nk = lib.raw("hific_prof_end")(MAXK, ms, fl, cnt, names)
tag = f"C{C} K{K} H{H} s{ST} " + " ".join(f"{k[6:]}={v}" for k, v in sorted(os.environ.items()) if k.startswith("HIFIC_") and k != "HIFIC_LIB_PATH")
for k in range(nk):
    nm = names.raw[k * 64:(k + 1) * 64].split(b"\0", 1)[0].decode()
    if cnt[k]:
        print(f"{tag} | {nm}: {ms[k] * 1e3 / cnt[k]:.1f} us x{cnt[k]}  {fl[k] / (ms[k] * 1e-3) / 1e12:.0f} TF/s", flush=True)

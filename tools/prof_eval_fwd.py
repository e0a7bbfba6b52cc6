"""Per-launch table of the EVALUATION-mode forward (bench.py's fwd_ms_per_image leg): HIFIC_PROF_DUMP=1 python tools/prof_eval_fwd.py 2> dump"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hific_amd
from hific_amd import lib
from hific_amd.default_config import make_args, hific_args, ModelTypes, ModelModes
dev = torch.device("cuda:0")
hific_amd.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
m = hific_amd.Model(make_args(hific_args, batch_size=16), model_type=ModelTypes.COMPRESSION_GAN, model_mode=ModelModes.EVALUATION,
                    allow_random_lpips_backbone=True, build_tables=False).to(dev).eval()
x = torch.rand((16, 3, 256, 256), device=dev)
with torch.no_grad():
    for _ in range(3): m(x)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): m(x)
    torch.cuda.synchronize(); print("fwd ms/image", (time.perf_counter() - t) / 10 / 16 * 1e3)
    lib.call("hific_prof_begin")
    for _ in range(2): m(x)
    MAXK = 32
    ms = (ctypes.c_double * MAXK)(); fl = (ctypes.c_double * MAXK)(); cnt = (ctypes.c_int * MAXK)(); names = ctypes.create_string_buffer(MAXK * 64)
    lib.raw("hific_prof_end")(MAXK, ms, fl, cnt, names)

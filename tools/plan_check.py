"""Host-side dry run of the conv planner for every layer geometry of a configuration (no GPU needed): the pack-plan
entry points run the forward / data-gradient planner without launching; the weight-gradient entry points are called with
dummy pointers (they fail at the launch on a box without a GPU, i.e. rc -3, but report planner errors -2 / -4 first).
Usage: python tools/plan_check.py BATCH SIZE [dtype: bf16|f32]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hific_amd import lib  # noqa: E402

B, S = int(sys.argv[1]), int(sys.argv[2])
cd = lib.HIFIC_F32 if (len(sys.argv) > 3 and sys.argv[3] == "f32") else lib.HIFIC_BF16
WS = int(os.environ.get("HIFIC_WS_MB", "1536")) << 20
R, Z = lib.PAD_REFLECT, lib.PAD_ZERO
convs, convts = [], []      # (name, N, C, H, W, K, R, stride, pads, mode) / (name, N, Ci, H, W, Co, R, stride, pad, outpad)
f = (60, 120, 240, 480, 960)
convs.append(("E1", B, 3, S, S, 60, 7, 1, (3, 3, 3, 3), R))
h = S
for i in range(4):
    convs.append((f"E{i + 2}", B, f[i], h, h, f[i + 1], 3, 2, (1, 0, 0, 1), R)); h //= 2
convs.append(("E6", B, 960, h, h, 220, 3, 1, (1, 1, 1, 1), R))
convs.append(("Ginit", B, 220, h, h, 960, 3, 1, (1, 1, 1, 1), R))
convs.append(("Rblock", B, 960, h, h, 960, 3, 1, (1, 1, 1, 1), R))
g = h
for i in range(4):
    convts.append((f"U{i + 1}", B, f[4 - i], g, g, f[3 - i], 3, 2, 1, 1)); g *= 2
convs.append(("G9", B, 60, S, S, 3, 7, 1, (3, 3, 3, 3), R))
convs.append(("A1", B, 220, h, h, 320, 3, 1, (1, 1, 1, 1), Z))
convs.append(("A2", B, 320, h, h, 320, 5, 2, (2, 2, 2, 2), R))
convs.append(("A3", B, 320, h // 2, h // 2, 320, 5, 2, (2, 2, 2, 2), R))
convts.append(("S1", B, 320, h // 4, h // 4, 320, 5, 2, 2, 1))
convts.append(("S2", B, 320, h // 2, h // 2, 320, 5, 2, 2, 1))
convts.append(("S3", B, 320, h, h, 220, 3, 1, 1, 0))
convs.append(("Dctx", 2 * B, 220, h, h, 12, 3, 1, (1, 1, 1, 1), R))
d, c = S, 15
for i, k in enumerate((64, 128, 256, 512)):
    convs.append((f"D{i + 1}", 2 * B, c, d, d, k, 4, 2, (1, 1, 1, 1), R)); d //= 2; c = k
convs.append(("Dout", 2 * B, 512, d, d, 1, 1, 1, (0, 0, 0, 0), Z))
a = (S + 4 - 11) // 4 + 1
convs.append(("L1", 2 * B, 3, S, S, 64, 11, 4, (2, 2, 2, 2), Z))
a2 = (a - 3) // 2 + 1
convs.append(("L2", 2 * B, 64, a2, a2, 192, 5, 1, (2, 2, 2, 2), Z))
a3 = (a2 - 3) // 2 + 1
for nm, ci, co in (("L3", 192, 384), ("L4", 384, 256), ("L5", 256, 256)):
    convs.append((nm, 2 * B, ci, a3, a3, co, 3, 1, (1, 1, 1, 1), Z))

job = ctypes.create_string_buffer(int(lib.raw("hific_pack_job_bytes")()))
dummy = ctypes.c_void_p(0x100000)
bad = 0
for nm, N, C, H, W, K, Rk, st, pads, mode in convs:
    ws = lib.raw("hific_conv2d_ws_bytes")(N, C, H, W, K, Rk, Rk, st, *pads, cd)
    rcs = [lib.raw("hific_conv2d_pack_plan")(kind, N, C, H, W, K, Rk, Rk, st, *pads, mode, cd, 0, job, len(job)) for kind in (0, 1)]
    rw = lib.raw("hific_conv2d_bwd_weight")(dummy, dummy, dummy, N, C, H, W, K, Rk, Rk, st, *pads, mode, 0, cd, 0, dummy, WS, None)
    ok = rcs == [0, 0] and rw in (0, -3) and ws <= WS
    bad += not ok
    print(f"{nm:7s} N{N} C{C} {H}x{W} K{K} r{Rk} s{st}: ws {ws / 2**20:8.1f} MiB  plan fwd/bwd {rcs}  wgrad rc {rw} {'' if ok else '<-- PROBLEM'}")
for nm, N, Ci, H, W, Co, Rk, st, pad, op in convts:
    ws = lib.raw("hific_conv_transpose2d_ws_bytes")(N, Ci, H, W, Co, Rk, Rk, st, pad, op, cd)
    rcs = [lib.raw("hific_conv_transpose2d_pack_plan")(kind, N, Ci, H, W, Co, Rk, Rk, st, pad, op, cd, 0, job, len(job)) for kind in (0, 1)]
    rw = lib.raw("hific_conv_transpose2d_bwd_weight")(dummy, dummy, dummy, N, Ci, H, W, Co, Rk, Rk, st, pad, op, 0, cd, 0, dummy, WS, None)
    ok = rcs == [0, 0] and rw in (0, -3) and ws <= WS
    bad += not ok
    print(f"{nm:7s} N{N} Ci{Ci} {H}x{W} Co{Co} r{Rk} s{st}: ws {ws / 2**20:8.1f} MiB  plan fwd/bwd {rcs}  wgrad rc {rw} {'' if ok else '<-- PROBLEM'}")
print("problems:", bad)

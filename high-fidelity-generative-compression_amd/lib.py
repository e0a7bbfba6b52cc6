"""ctypes binding of libhific_hip.so (the C-ABI declared in include/hific_hip.h).

The product path has NO fallback: if the shared library is missing or a symbol is absent this module raises,
and every op raises when handed CPU tensors.  PyTorch is used only for device memory, streams and autograd
bookkeeping; all arithmetic of the hot path runs inside the library.
"""
import ctypes
import os
from ctypes import c_int, c_float, c_void_p, c_size_t, c_longlong, c_char_p, POINTER

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HIFIC_LIB_PATH") or os.path.join(_HERE, "libhific_hip.so")   # (override: A/B builds of the library)

HIFIC_F32, HIFIC_BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_LEAKY = 0, 1, 2
PAD_ZERO, PAD_REFLECT = 0, 1

_ERR = {0: "ok", -1: "bad argument", -2: "workspace too small", -3: "kernel launch failed", -4: "unsupported shape"}

P, I, F, Z, L = c_void_p, c_int, c_float, c_size_t, c_longlong

# name -> (restype, argtypes)   (kept in the same order as include/hific_hip.h)
SIGNATURES = {
    "hific_version": (I, []),
    "hific_device_info": (I, [I, c_char_p, POINTER(c_int), POINTER(c_int)]),
    "hific_set_ticket_buffer": (I, [P, P, Z]),
    "hific_conv2d_ws_bytes": (Z, [I] * 13),
    "hific_conv_transpose2d_ws_bytes": (Z, [I] * 11),
    "hific_conv2d_fwd": (I, [P, P, P, P, P, P] + [I] * 16 + [P, Z, P, Z, I, P]),
    "hific_conv2d_bwd_data": (I, [P, P, P, P] + [I] * 15 + [P, Z, P, Z, I, P]),
    "hific_conv2d_bwd_weight": (I, [P, P, P] + [I] * 16 + [P, Z, P]),
    "hific_conv_transpose2d_fwd": (I, [P, P, P, P] + [I] * 13 + [P, Z, P, Z, I, P]),
    "hific_conv_transpose2d_bwd_data": (I, [P, P, P] + [I] * 12 + [P, Z, P, Z, I, P]),
    "hific_conv_transpose2d_bwd_weight": (I, [P, P, P] + [I] * 13 + [P, Z, P]),
    "hific_channelnorm_fwd": (I, [P, P, P, P, P, P, I, I, I, F, I, I, P]),
    "hific_channelnorm_fwd_res": (I, [P, P, P, P, P, P, P, I, I, I, F, I, I, P]),
    "hific_channelnorm_fwd_exact": (I, [P, P, P, P, P, P, P, P, I, I, I, F, I, I, P]),
    "hific_channelnorm_fwd_exact_res": (I, [P, P, P, P, I, P, P, P, P, P, I, I, I, F, I, I, P]),
    "hific_channelnorm_bwd_ws_bytes": (Z, [I, I, I]),
    "hific_channelnorm_bwd": (I, [P] * 9 + [I, I, I, I, I, I, P, Z, P, I, P]),
    "hific_act_bwd": (I, [P, P, P, L, F, I, P]),
    "hific_tanh_fwd": (I, [P, P, L, I, P]),
    "hific_tanh_bwd": (I, [P, P, P, L, I, P]),
    "hific_scale_shift": (I, [P, P, L, F, F, I, P]),
    "hific_add": (I, [P, P, P, L, I, P]),
    "hific_cast": (I, [P, I, P, I, L, P]),
    "hific_pad2d": (I, [P, P, L, I, I, I, I, I, I, I, I, P]),
    "hific_add_split": (I, [P, I, P, I, P, P, I, I, I, I, P]),
    "hific_split3": (I, [P, P, L, I, L, I, I, P]),
    "hific_axpby_f32": (I, [P, P, P, F, F, L, P]),
    "hific_channel_sum": (I, [P, P, I, I, I, I, I, P, Z, P]),
    "hific_maxpool2s2_fwd": (I, [P, P, L, I, I, I, P]),
    "hific_maxpool2s2_bwd": (I, [P, P, P, L, I, I, I, P]),
    "hific_maxpool3s2_fwd": (I, [P, P, L, I, I, I, P]),
    "hific_maxpool3s2_bwd": (I, [P, P, P, L, I, I, I, P]),
    "hific_loss_combine_fwd": (I, [P, P, I, P, P, P, F, F, F, F, F, F, P, P, P]),
    "hific_loss_combine_bwd": (I, [P, P, I, F, F, F, P, P]),
    "hific_mse_fwd": (I, [P, P, P, L, F, I, P, Z, P]),
    "hific_mse_bwd": (I, [P, P, P, P, L, F, I, P]),
    "hific_bce_fwd": (I, [P, F, P, L, P, Z, P]),
    "hific_bce_bwd": (I, [P, F, P, P, L, I, P]),
    "hific_lsq_sigmoid_fwd": (I, [P, F, P, L, P, Z, P]),
    "hific_lsq_sigmoid_bwd": (I, [P, F, P, P, L, I, P]),
    "hific_sigmoid_f32": (I, [P, P, L, P]),
    "hific_upcat_fwd": (I, [P, P, P, I, I, I, I, I, I, I, P]),
    "hific_upcat_bwd": (I, [P, P, I, I, P, I, I, I, I, I, I, I, P]),
    "hific_upcat_pair_fwd": (I, [P, P, P, P] + [I] * 7 + [P]),
    "hific_upcat_pair_bwd": (I, [P, P, P] + [I] * 7 + [P]),
    "hific_d1_ctx_grad": (I, [P, P, P, P, I, I, I, I, I, I, I, I, P, Z, P]),
    "hific_spectral_norm_fwd": (I, [P, P, P, P, I, I, I, F, P, Z, P]),
    "hific_spectral_norm_fwd_batch": (I, [POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p),
                                          POINTER(c_int), POINTER(c_int), I, I, F, P, Z, P]),
    "hific_spectral_norm_bwd": (I, [P, P, P, P, P, P, I, I, I, P, Z, P]),
    "hific_adam_step": (I, [P, P, P, P, L, F, F, F, F, I, F, P]),
    "hific_adam_prepare": (I, [P, P, F, F, P]),
    "hific_adam_apply": (I, [P, P, P, P, L, F, F, F, F, P, F, P]),
    "hific_prior_symbols": (I, [P, P, P, P, I, F, P, P, L, P]),
    "hific_hyper_symbols": (I, [P, P, P, I, I, I, P]),
    "hific_round_f32": (I, [P, P, P, L, P]),
    "hific_lower_bound_fwd": (I, [P, F, P, L, P]),
    "hific_lower_bound_bwd": (I, [P, P, F, P, L, P]),
    "hific_logsum_fwd": (I, [P, P, L, F, F, P, Z, P]),
    "hific_logsum_bwd": (I, [P, P, P, L, F, F, I, P]),
    "hific_gauss_lik_fwd": (I, [P, P, P, P, L, F, I, P]),
    "hific_gauss_lik_bwd": (I, [P, P, P, P, P, P, P, L, F, I, I, I, P]),
    "hific_factorized_lik_fwd": (I, [P, POINTER(c_void_p), P, I, I, I, F, P]),
    "hific_factorized_lik_bwd": (I, [P, POINTER(c_void_p), P, P, POINTER(c_void_p), I, I, I, F, I, P, Z, P]),
    "hific_lpips_prep": (I, [P, I, P, I, P, I, I, I, I, P]),
    "hific_lpips_prep_bwd": (I, [P, P, I, I, I, I, I, P]),
    "hific_lpips_tap_fwd": (I, [P, P, P, I, I, I, I, I, P, Z, P]),
    "hific_lpips_tap_bwd": (I, [P, P, P, P, I, I, I, I, I, P]),
    "hific_pack_job_bytes": (Z, []),
    "hific_conv2d_pack_plan": (I, [I] * 16 + [P, Z]),
    "hific_conv_transpose2d_pack_plan": (I, [I] * 13 + [P, Z]),
    "hific_pack_job_set_ptrs": (I, [P, P, P, P]),
    "hific_pack_job_info": (I, [P, POINTER(c_int), POINTER(c_int), POINTER(c_longlong), POINTER(c_int)]),
    "hific_pack_batch": (I, [P, P, I, I, Z, I, P]),
    "hific_augment_crop": (I, [P, P, P, P, P, I, I, I, I, P, P]),
    "hific_prof_begin": (I, []),
    "hific_env_refresh": (I, []),
    "hific_prof_end": (I, [I, POINTER(ctypes.c_double), POINTER(ctypes.c_double), POINTER(c_int), c_char_p]),
    "hific_prof_bytes": (I, [I, POINTER(ctypes.c_double)]),
}


class HificError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"(hipcc --offload-arch=gfx950). There is no CPU/PyTorch fallback for the HiFIC hot path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing: loud by design
        fn.restype = res
        fn.argtypes = args
    return lib


_lib = _load()


def raw(name):
    return getattr(_lib, name)


def call(name, *args):
    rc = getattr(_lib, name)(*args)
    if rc != 0:
        raise HificError(f"{name} failed: {_ERR.get(rc, rc)} (rc={rc})")


# ---- torch-side plumbing (device memory, stream) -----------------------------------------------------
import torch  # noqa: E402

_workspaces = {}
_WS_BYTES = int(os.environ.get("HIFIC_WS_MB", "1536")) << 20


def dtype_code(t):
    if t.dtype == torch.float32:
        return HIFIC_F32
    if t.dtype == torch.bfloat16:
        return HIFIC_BF16
    raise HificError(f"unsupported tensor dtype {t.dtype}")


def require_gpu(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise HificError("hific_amd ops need GPU (ROCm) tensors: the HIP library is the only implementation "
                             "of the hot path (no CPU fallback)")
        if not t.is_contiguous():
            raise HificError("hific_amd ops need contiguous NCHW tensors")


def ptr(t):
    return None if t is None else t.data_ptr()


# torch.cuda.current_stream() costs ~8 us per call (device-index normalisation through is_available()); the raw accessors are
# two C calls.  ~2000 stream queries per GAN cycle: 4-5 ms of host time per cycle.
_raw_stream = torch._C._cuda_getCurrentRawStream
_cur_device = torch._C._cuda_getDevice


def stream():
    return _raw_stream(_cur_device())


_stream_objs = {}


def stream_obj(device=None):
    """torch.cuda.current_stream(device) without its ~8 us of device-index normalisation: Stream objects cached by raw
    handle (the current stream of the CURRENT device; callers pass `device` only for symmetry - ops run under the tensor's
    device).  ~150 queries per training cycle."""
    dev = _cur_device()
    raw = _raw_stream(dev)
    key = (dev, raw)
    st = _stream_objs.get(key)
    if st is None:
        st = _stream_objs[key] = torch.cuda.current_stream(dev)
    return st


def workspace(device, min_bytes=0):
    """Persistent per-device scratch (packed weights, split-K partials, padded-grad buffers).  Re-used by every op:
    safe because all ops are stream-ordered on the current stream."""
    cur = _cur_device()
    key = (device.index if device.index is not None else cur, _raw_stream(cur))
    ws = _workspaces.get(key)
    need = max(_WS_BYTES, min_bytes)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=device)
        _workspaces[key] = ws
        if key not in _tickets and _TICKETS_ON:
            # the stream's ticket counters (include/hific_hip.h "tickets"): zeroed once, self-cleaning afterwards
            tk = torch.zeros(_TICKET_BYTES, dtype=torch.uint8, device=device)
            _tickets[key] = tk
            with torch.cuda.device(key[0]):                  # the registry is keyed by (current device, stream)
                call("hific_set_ticket_buffer", key[1], tk.data_ptr(), tk.numel())
    return ws


_tickets = {}
_TICKET_BYTES = 16384
_TICKETS_ON = os.environ.get("HIFIC_TICKETS", "1") != "0"


def set_tickets(on):
    """Registers / removes the ticket buffers of every stream that has a workspace (tests: A/B of the two-launch forms)."""
    global _TICKETS_ON
    _TICKETS_ON = bool(on)
    for key, ws in _workspaces.items():
        with torch.cuda.device(key[0]):
            if on:
                tk = _tickets.get(key)
                if tk is None:
                    tk = _tickets[key] = torch.zeros(_TICKET_BYTES, dtype=torch.uint8, device=ws.device)
                call("hific_set_ticket_buffer", key[1], tk.data_ptr(), tk.numel())
            else:
                call("hific_set_ticket_buffer", key[1], None, 0)


def exported_symbols():
    return sorted(SIGNATURES)

"""Host rANS coder of the EVALUATION path: bit-compatible native port (libhific_host.so,
include/hific_host.h) of the reference's `compression_utils.ans_compress / ans_decompress`
(src/compression/compression_utils.py:130-229 -> entropy_coding.py:107-268, 271-476, 479-673 over ans.py:45-96).

Same call signatures and return conventions as the reference:
  ans_compress(symbols, indices, cdf, cdf_length, cdf_offset, coding_shape, precision, vectorize, block_encode)
      -> (encoded uint32 ndarray, coding_shape)           [block_encode=False, scalar: list of such pairs]
  ans_decompress(encoded, indices, cdf, cdf_length, cdf_offset, coding_shape, precision, vectorize, block_decode)
      -> decoded symbols, int32, shape of `indices`
`symbols` / `indices` are the int32 (N,C,H,W) tensors that `ops.prior_symbols_and_indices` /
`ops.hyper_symbols_and_indices` produce on the device (numpy arrays or torch tensors are accepted).
"""
import ctypes

import numpy as np
import torch

from . import tables as _tables

PATCH_SIZE = (1, 1)            # entropy_coding.py:10


class RansError(RuntimeError):
    pass


_ERR = {-1: "invalid argument", -6: "index / cdf_length / precision out of range, or a zero-width interval",
        -7: "output buffer too small", -8: "corrupt or truncated message"}
_bound = False


def _lib():
    global _bound
    lib = _tables._load()
    if not _bound:
        P, LL, I = ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int
        lib.hific_rans_encode.restype = I
        lib.hific_rans_encode.argtypes = [P, P, LL, P, I, I, P, P, I, P, LL, P]
        lib.hific_rans_decode.restype = I
        lib.hific_rans_decode.argtypes = [P, LL, P, LL, P, I, I, P, P, I, P]
        lib.hific_rans_encode_vec.restype = I
        lib.hific_rans_encode_vec.argtypes = [P, P, LL, LL, P, I, I, P, P, I, P, LL, P]
        lib.hific_rans_decode_vec.restype = I
        lib.hific_rans_decode_vec.argtypes = [P, LL, P, LL, LL, P, I, I, P, P, I, P]
        _bound = True
    return lib


def _np(a, dtype):
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    return np.ascontiguousarray(np.asarray(a).astype(dtype, copy=False))


def _tables_np(cdf, cdf_length, cdf_offset):
    cdf = _np(cdf, np.uint32)
    if cdf.ndim != 2:
        raise RansError("cdf must be [rows, max_length + 2]")
    return cdf, _np(cdf_length, np.int32), _np(cdf_offset, np.int32)


def _check(rc, what):
    if rc:
        raise RansError(f"{what}: {_ERR.get(rc, rc)}")


def _steps_layout(x):
    """The reference's vectorisation: batch 1 -> one step per (h, w) patch with the C channels as lanes
    (`compression_utils.decompose`, PATCH_SIZE (1,1)); batch > 1 -> one step per batch element, C*H*W lanes.
    -> (int32 [steps][lanes] ndarray, steps, lanes).  A tensor that lives on the GPU (what the device symbol kernels
    hand over) is laid out there, before the copy to the host: the (C, H*W) -> (H*W, C) transpose of a megapixel
    image's 900k symbols costs the host 1.2 ms per tensor and the device nothing."""
    B, C, H, W = x.shape
    if isinstance(x, torch.Tensor):
        x = x.detach().to(torch.int32)
        y = x[0].reshape(C, H * W).t().contiguous() if B == 1 else x.reshape(B, -1).contiguous()
        y = y.cpu().numpy()
    elif B == 1:
        y = np.ascontiguousarray(x[0].reshape(C, H * W).T)
    else:
        y = np.ascontiguousarray(x.reshape(B, -1))
    return (y, H * W, C) if B == 1 else (y, B, C * H * W)


def _sym_idx(symbols, indices):
    """Shape checks without forcing device tensors through the host twice."""
    if tuple(symbols.shape) != tuple(indices.shape) or len(symbols.shape) != 4:
        raise RansError("symbols and indices must be equally shaped (N,C,H,W) tensors")
    f = lambda a: a if isinstance(a, torch.Tensor) and a.is_cuda else _np(a, np.int32)
    return f(symbols), f(indices)


def _encode_scalar(sym, idx, cdf, cl, co, precision):
    lib = _lib()
    sym, idx = np.ascontiguousarray(sym.ravel()), np.ascontiguousarray(idx.ravel())
    need = ctypes.c_longlong(0)
    args = (sym.ctypes.data, idx.ctypes.data, sym.size, cdf.ctypes.data, cdf.shape[0], cdf.shape[1], cl.ctypes.data,
            co.ctypes.data, int(precision))
    out = np.empty(2 + (sym.size * int(precision) + 31) // 32 + sym.size // 8 + 1024, dtype=np.uint32)   # see ans_compress
    rc = lib.hific_rans_encode(*args, out.ctypes.data, out.size, ctypes.byref(need))
    if rc == -7:
        out = np.empty(need.value, dtype=np.uint32)
        rc = lib.hific_rans_encode(*args, out.ctypes.data, out.size, ctypes.byref(need))
    _check(rc, "hific_rans_encode")
    return out[:need.value].copy()


def ans_compress(symbols, indices, cdf, cdf_length, cdf_offset, coding_shape, precision, vectorize=False,
                 block_encode=True):
    sym, idx = _sym_idx(symbols, indices)
    cdf, cl, co = _tables_np(cdf, cdf_length, cdf_offset)
    if vectorize:
        lib = _lib()
        s, T, L = _steps_layout(sym)
        i, _, _ = _steps_layout(idx)
        need = ctypes.c_longlong(0)
        args = (s.ctypes.data, i.ctypes.data, T, L, cdf.ctypes.data, cdf.shape[0], cdf.shape[1], cl.ctypes.data,
                co.ctypes.data, int(precision))
        # one pass in the usual case: a symbol costs at most `precision` bits of 32-bit words, overflow nibbles are rare;
        # the coder reports the exact size when the guess is too small (-7) and is then called again
        out = np.empty(2 * L + (T * L * int(precision) + 31) // 32 + T * L // 8 + 1024, dtype=np.uint32)
        rc = lib.hific_rans_encode_vec(*args, out.ctypes.data, out.size, ctypes.byref(need))
        if rc == -7:
            out = np.empty(need.value, dtype=np.uint32)
            rc = lib.hific_rans_encode_vec(*args, out.ctypes.data, out.size, ctypes.byref(need))
        _check(rc, "hific_rans_encode_vec")
        out = out[:need.value].copy()
        shape = (int(sym.shape[1]),) + PATCH_SIZE if sym.shape[0] == 1 else tuple(coding_shape)
        return out, shape
    sym, idx = _np(sym, np.int32), _np(idx, np.int32)
    if block_encode:
        return _encode_scalar(sym, idx, cdf, cl, co, precision), tuple(sym.shape[1:])
    return [(_encode_scalar(sym[b], idx[b], cdf, cl, co, precision), tuple(sym.shape[2:])) for b in range(sym.shape[0])]


def ans_decompress(encoded, indices, cdf, cdf_length, cdf_offset, coding_shape, precision, vectorize=False,
                   block_decode=True, device=None):
    """`device` (extension, vectorised path): return the symbols as an int32 torch tensor on that device instead of a
    numpy array - the [steps][lanes] -> (N,C,H,W) transpose then happens after the upload."""
    cdf, cl, co = _tables_np(cdf, cdf_length, cdf_offset)
    lib = _lib()
    if vectorize:
        enc = _np(encoded, np.uint32)
        if len(indices.shape) != 4:
            raise RansError("indices must be an (N,C,H,W) tensor")
        B, C, H, W = (int(d) for d in indices.shape)
        i, T, L = _steps_layout(indices if isinstance(indices, torch.Tensor) and indices.is_cuda else _np(indices, np.int32))
        out = np.empty(T * L, dtype=np.int32)
        _check(lib.hific_rans_decode_vec(enc.ctypes.data, enc.size, i.ctypes.data, T, L, cdf.ctypes.data, cdf.shape[0],
                                         cdf.shape[1], cl.ctypes.data, co.ctypes.data, int(precision),
                                         out.ctypes.data), "hific_rans_decode_vec")
        if device is not None:
            t = torch.from_numpy(out).to(device)
            return t.reshape(H * W, C).t().reshape(1, C, H, W).contiguous() if B == 1 else t.reshape(B, C, H, W)
        return out.reshape(H * W, C).T.reshape(1, C, H, W).copy() if B == 1 else out.reshape(B, C, H, W)
    idx = _np(indices, np.int32)

    def scalar(enc, ind):
        enc = _np(enc, np.uint32)
        flat = np.ascontiguousarray(ind.ravel())
        out = np.empty(flat.size, dtype=np.int32)
        _check(lib.hific_rans_decode(enc.ctypes.data, enc.size, flat.ctypes.data, flat.size, cdf.ctypes.data,
                                     cdf.shape[0], cdf.shape[1], cl.ctypes.data, co.ctypes.data, int(precision),
                                     out.ctypes.data), "hific_rans_decode")
        return out.reshape(ind.shape)

    if block_decode:
        return scalar(encoded, idx)
    return np.stack([scalar(encoded[b][0] if isinstance(encoded[b], tuple) else encoded[b], idx[b])
                     for b in range(idx.shape[0])], axis=0)

"""Test infrastructure (not product code): the two shims that let the reference's rANS codec run in this image, so
that `hific_amd.compression.rans` can be pinned against it byte for byte.

1. NumPy 2 (NEP 50): `ans.py` computes `((RANS_L >> precision) << 32) * freqs` with `RANS_L` a Python int and `freqs`
   a uint32 array; NumPy 1 promoted that to uint64 by value, NumPy 2 raises OverflowError.  Making `RANS_L` an
   `np.uint64` restores the arithmetic the reference was written against (same values, same dtype as under NumPy 1).
2. `compression_utils.view_update` is built on the `autograd` package, which is not installed (no network).  For the
   only views the codec uses (boolean-mask indexing of the coder heads, `entropy_coding.overflow_view`) it means
   "return head[mask] and a function that writes a new value back into a copy": restated with plain NumPy.
"""
import numpy as np


def apply():
    from src.compression import ans as vrans, compression_utils

    vrans.RANS_L = np.uint64(1 << 31)

    def view_update(data, view_fun):
        pos = view_fun(np.arange(data.size).reshape(data.shape))
        item = view_fun(data)

        def update(new_item):
            out = np.array(data, copy=True)
            out.flat[pos] = new_item
            return out
        return item, update

    compression_utils.view_update = view_update
    return vrans, compression_utils

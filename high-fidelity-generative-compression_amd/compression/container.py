"""`.hfc` container of the EVALUATION path: byte-compatible restatement of
`compression_utils.save_compressed_format / load_compressed_format` (src/compression/compression_utils.py:300-371).

Layout (little endian): uint16 hyperlatent (H,W) | uint16 image (H,W) | uint16 hyper coding shape (C,H,W) |
uint16 latent coding shape (C,H,W) | uint16 batch | MAGIC | uint32 nbytes | hyperlatent message (uint32 words) |
MAGIC | uint32 nbytes | latent message | MAGIC,  MAGIC = 46 E2 84 92.
"""
import os
import struct
from collections import namedtuple

import numpy as np

MAGIC = b"\x46\xE2\x84\x92"                       # compression_utils.py:16

CompressionOutput = namedtuple("CompressionOutput",             # compression_utils.py:20-28
                               ["hyperlatents_encoded", "latents_encoded", "hyperlatent_spatial_shape",
                                "batch_shape", "spatial_shape", "hyper_coding_shape", "latent_coding_shape"])


class ContainerError(RuntimeError):
    pass


def _u16s(values, n):
    values = [int(v) for v in values]
    if len(values) != n or any(v < 0 or v >= 2 ** 16 for v in values):
        raise ContainerError(f"expected {n} values below 2**16, got {values}")
    return struct.pack(f"<{n}H", *values)


def save_compressed_format(compression_output, out_path):
    """Writes the container; returns (actual_bpp, theoretical_bpp) like the reference (theoretical = `total_bpp` of
    the output when it has one, else NaN)."""
    co = compression_output
    hyp = np.ascontiguousarray(np.asarray(co.hyperlatents_encoded, dtype=np.uint32))
    lat = np.ascontiguousarray(np.asarray(co.latents_encoded, dtype=np.uint32))
    if hyp.nbytes >= 2 ** 32 or lat.nbytes >= 2 ** 32:
        raise ContainerError("message longer than 2**32 bytes")
    with open(out_path, "wb") as f:
        f.write(_u16s(co.hyperlatent_spatial_shape, 2))
        f.write(_u16s(co.spatial_shape, 2))
        f.write(_u16s(co.hyper_coding_shape, 3))
        f.write(_u16s(co.latent_coding_shape, 3))
        f.write(_u16s([co.batch_shape], 1))
        f.write(MAGIC)
        f.write(struct.pack("<I", hyp.nbytes)); f.write(hyp.tobytes()); f.write(MAGIC)
        f.write(struct.pack("<I", lat.nbytes)); f.write(lat.tobytes()); f.write(MAGIC)
    actual_bpp = 8.0 * float(os.path.getsize(out_path)) / float(np.prod(co.spatial_shape))
    total = getattr(co, "total_bpp", float("nan"))
    return actual_bpp, float(total.item() if hasattr(total, "item") else total)


def load_compressed_format(in_path):
    with open(in_path, "rb") as f:
        def u16s(n):
            raw = f.read(2 * n)
            if len(raw) != 2 * n:
                raise ContainerError("truncated header")
            return tuple(int(v) for v in struct.unpack(f"<{n}H", raw))

        def fence():
            if f.read(4) != MAGIC:
                raise ContainerError("not an .hfc file (separator missing)")

        def message():
            raw = f.read(4)
            if len(raw) != 4:
                raise ContainerError("truncated message header")
            n = struct.unpack("<I", raw)[0]
            data = f.read(n)
            if len(data) != n or n % 4:
                raise ContainerError("truncated message")
            return np.frombuffer(data, dtype=np.uint32).copy()

        hyper_hw = u16s(2); image_hw = u16s(2); hyper_shape = u16s(3); latent_shape = u16s(3); batch = u16s(1)[0]
        fence()
        hyp = message(); fence()
        lat = message(); fence()
    return CompressionOutput(hyperlatents_encoded=hyp, latents_encoded=lat, hyperlatent_spatial_shape=hyper_hw,
                             batch_shape=batch, spatial_shape=image_hw, hyper_coding_shape=hyper_shape,
                             latent_coding_shape=latent_shape)

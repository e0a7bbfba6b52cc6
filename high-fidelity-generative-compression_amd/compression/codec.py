"""EVALUATION path: `Hyperprior.compress_forward / decompress_forward` (src/hyperprior.py:195-274) as a
device-agnostic orchestration of the pieces this package provides:

    networks (analysis / synthesis_mu / synthesis_std)  - the HIP modules on the GPU, or any callables
    symbol extraction                                    - ops.prior_symbols_and_indices / hyper_symbols_and_indices
    rANS coder + tables                                  - compression.rans / compression.tables (libhific_host.so)

The networks and the symbol extractors are passed in as callables, so the same code runs with the device modules
(`hific_amd.hyperprior.Hyperprior.compress_forward`) and, in the CPU tests, with the oracle's functional networks,
where it is pinned byte for byte against the reference's `compress_forward`.
"""
from collections import namedtuple

import numpy as np
import torch

from . import rans

# what `Hyperprior.compress_forward` returns (src/hyperprior.py:25-39): the 7 fields the container stores plus the
# Shannon estimates that Model.compress / compress.py / save_compressed_format report (`.total_bpp` etc.).  The loader
# (container.load_compressed_format) returns the reference's 7-field tuple of compression_utils.py:20-28.
CompressionOutput = namedtuple("CompressionOutput",
                               ["hyperlatents_encoded", "latents_encoded", "hyperlatent_spatial_shape", "batch_shape",
                                "spatial_shape", "hyper_coding_shape", "latent_coding_shape", "hyperlatent_bits",
                                "latent_bits", "total_bits", "hyperlatent_bpp", "latent_bpp", "total_bpp"])

EntropyTables = namedtuple("EntropyTables", ["CDF", "CDF_offset", "CDF_length"])          # int32 tensors / arrays
CodecNets = namedtuple("CodecNets", ["analysis", "synthesis_mu", "synthesis_std"])        # callables tensor -> tensor
SymbolFns = namedtuple("SymbolFns", ["hyper", "prior", "prior_indices"])
# hyper(z) -> (symbols, indices); prior(y, means, scales, scale_table) -> (symbols, indices);
# prior_indices(scales, scale_table) -> indices          (all int32, (N,C,H,W))

PRECISION = 16                 # entropy_models.PRECISION_P
SCALE_LOWER_BOUND = 0.11       # hyperprior.py: scale_lower_bound / MIN_SCALE


def _tab(t):
    f = lambda a: a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    return f(t.CDF).astype(np.uint32), f(t.CDF_length).astype(np.int32), f(t.CDF_offset).astype(np.int32)


def _decoded(encoded, idx, cdf, cl, co, coding_shape, precision, vectorize, block, device):
    """rANS-decode to a tensor on `device`.  The vectorised coder hands back its [steps][lanes] array and the layout change
    to (N,C,H,W) is done where the tensor is going to live (rans.ans_decompress `device`)."""
    if vectorize:
        return rans.ans_decompress(encoded, idx, cdf, cl, co, coding_shape, precision, vectorize=True, block_decode=block,
                                   device=device)
    idx = idx.detach().cpu().numpy() if isinstance(idx, torch.Tensor) else np.asarray(idx)
    sym = rans.ans_decompress(encoded, idx, cdf, cl, co, coding_shape, precision, vectorize=False, block_decode=block)
    return torch.from_numpy(np.asarray(sym)).to(device)


def compress_forward(latents, spatial_shape, nets, hyper_tables, prior_tables, scale_table, symbol_fns,
                     vectorize=True, block_encode=True, precision=PRECISION, scale_lower_bound=SCALE_LOWER_BOUND,
                     bits_fn=None):
    """hyperprior.py:195-246.  Returns the reference's 13-field `CompressionOutput`.  `bits_fn(hyperlatents, latents,
    means, scales) -> (hyperlatent_bits, latent_bits)` supplies the Shannon estimates of
    `_estimate_compression_bits` (hyperprior_model.py:108-131, prior_model.py:122-145); without it the reporting
    fields carry the ATTAINED sizes (32 bits per emitted rANS word)."""
    hyperlatents = nets.analysis(latents)
    hyper_hw = tuple(int(s) for s in hyperlatents.shape[2:])
    batch = int(latents.shape[0])
    # hyperlatents: symbols = floor(z + .5), one table row per channel (hyperprior_model.py:141-199)
    sym, idx = symbol_fns.hyper(hyperlatents)
    cdf, cl, co = _tab(hyper_tables)
    hyp_enc, hyper_coding_shape = rans.ans_compress(sym, idx, cdf, cl, co, tuple(sym.shape[1:]), precision,
                                                    vectorize=vectorize, block_encode=block_encode)
    # the encoder continues from what the decoder will see (hyperprior.py:211-215)
    hyperlatents_decoded = _decoded(hyp_enc, idx, cdf, cl, co, hyper_coding_shape, precision, vectorize, block_encode,
                                    latents.device).to(latents.dtype)
    means = nets.synthesis_mu(hyperlatents_decoded)
    scales = torch.clamp(nets.synthesis_std(hyperlatents_decoded), min=scale_lower_bound)    # LowerBoundToward fwd
    sym, idx = symbol_fns.prior(latents, means, scales, scale_table)
    cdf, cl, co = _tab(prior_tables)
    lat_enc, latent_coding_shape = rans.ans_compress(sym, idx, cdf, cl, co, tuple(sym.shape[1:]), precision,
                                                     vectorize=vectorize, block_encode=block_encode)
    if bits_fn is not None:
        hyper_bits, latent_bits = (float(b) for b in bits_fn(hyperlatents, latents, means, scales))
    else:
        hyper_bits, latent_bits = 32.0 * len(hyp_enc), 32.0 * len(lat_enc)
    n_pixels = float(np.prod([int(s) for s in spatial_shape]))
    return CompressionOutput(hyperlatents_encoded=hyp_enc, latents_encoded=lat_enc,
                             hyperlatent_spatial_shape=hyper_hw, batch_shape=batch,
                             spatial_shape=tuple(int(s) for s in spatial_shape),
                             hyper_coding_shape=tuple(int(s) for s in hyper_coding_shape),
                             latent_coding_shape=tuple(int(s) for s in latent_coding_shape),
                             hyperlatent_bits=hyper_bits, latent_bits=latent_bits, total_bits=hyper_bits + latent_bits,
                             hyperlatent_bpp=hyper_bits / n_pixels, latent_bpp=latent_bits / n_pixels,
                             total_bpp=(hyper_bits + latent_bits) / n_pixels)


def decompress_forward(compression_output, nets, hyper_tables, prior_tables, scale_table, symbol_fns, n_hyper_channels,
                       device="cpu", dtype=torch.float32, vectorize=True, block_decode=True, precision=PRECISION,
                       scale_lower_bound=SCALE_LOWER_BOUND):
    """hyperprior.py:248-274: returns the dequantised latents (symbols + means), (N,C,H,W) on `device`."""
    co_ = compression_output
    B = int(co_.batch_shape)
    Hh, Wh = (int(s) for s in co_.hyperlatent_spatial_shape)
    idx = np.ascontiguousarray(np.broadcast_to(np.arange(n_hyper_channels, dtype=np.int32).reshape(1, -1, 1, 1),
                                               (B, n_hyper_channels, Hh, Wh)))                # hyperprior_model.py:135-139
    cdf, cl, co = _tab(hyper_tables)
    hyperlatents_decoded = _decoded(co_.hyperlatents_encoded, idx, cdf, cl, co, tuple(co_.hyper_coding_shape), precision,
                                    vectorize, block_decode, device).to(dtype)
    means = nets.synthesis_mu(hyperlatents_decoded)
    scales = torch.clamp(nets.synthesis_std(hyperlatents_decoded), min=scale_lower_bound)
    idx = symbol_fns.prior_indices(scales, scale_table)
    cdf, cl, co = _tab(prior_tables)
    symbols = _decoded(co_.latents_encoded, idx, cdf, cl, co, tuple(co_.latent_coding_shape), precision, vectorize,
                       block_decode, means.device).to(means.dtype)
    return symbols + means                                                                     # dequantize, :246

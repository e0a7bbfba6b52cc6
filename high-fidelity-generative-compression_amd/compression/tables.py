"""Entropy-table construction for the EVALUATION path on the host.

`pmf_to_quantized_cdf` replaces the reference's O(n^2) pure-Python `src/helpers/maths.py:5-73` with the C++ port in
`libhific_host.so` (`include/hific_host.h`), bit-for-bit; `build_prior_tables` / `build_hyperprior_tables` mirror the
table loops of `src/compression/prior_model.py:77-120` and `src/compression/hyperprior_model.py:42-105` and return the
same `(CDF, CDF_offset, CDF_length)` int32 tensors the reference registers as parameters.  The probability mass
functions themselves are still evaluated with torch (same ops as the reference, so the float32 inputs to the
quantiser are identical); only the quantiser and the per-row loop are native.
"""
import ctypes
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(os.path.dirname(_HERE), "libhific_host.so")
_lib = None


class HostTablesError(RuntimeError):
    pass


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise HostTablesError(f"{_LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
        lib = ctypes.CDLL(_LIB_PATH)
        lib.hific_pmf_to_quantized_cdf.restype = ctypes.c_int
        lib.hific_pmf_to_quantized_cdf.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        lib.hific_build_cdf_rows.restype = ctypes.c_int
        lib.hific_build_cdf_rows.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                             ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
        lib.hific_host_version.restype = ctypes.c_char_p
        _lib = lib
    return _lib


_ERR = {-1: "invalid argument (precision outside [8,32], fewer than 2 entries, negative/NaN mass or zero total)",
        -5: "no symbol with frequency > 1 left to steal from"}


def pmf_to_quantized_cdf(pmf, precision):
    """maths.py:5-73: float32 pmf (n,) -> int64 cdf (n+1,), cdf[0] = 0, cdf[-1] = 2**precision."""
    lib = _load()
    p = pmf.detach().to(torch.float32).contiguous().cpu()
    if p.dim() != 1:
        raise HostTablesError("pmf must be one-dimensional")
    out = torch.empty(p.numel() + 1, dtype=torch.int64)
    rc = lib.hific_pmf_to_quantized_cdf(p.data_ptr(), p.numel(), int(precision), out.data_ptr())
    if rc:
        raise HostTablesError(f"hific_pmf_to_quantized_cdf: {_ERR.get(rc, rc)}")
    return out


def build_cdf_rows(pmf, lengths, extra, precision):
    """Per-row loop of `build_tables`: pmf (rows, max_len) float32, lengths (rows,) int32, extra (rows,) float32 mass
    appended to each row -> CDF (rows, max_len + 2) int32, rows zero-padded."""
    lib = _load()
    p = pmf.detach().to(torch.float32).contiguous().cpu()
    ln = lengths.detach().to(torch.int32).contiguous().cpu()
    ex = extra.detach().to(torch.float32).contiguous().cpu()
    rows, stride = p.shape
    width = int(ln.max().item()) + 2
    cdf = torch.zeros((rows, width), dtype=torch.int32)
    rc = lib.hific_build_cdf_rows(p.data_ptr(), rows, stride, ln.data_ptr(), ex.data_ptr(), int(precision),
                                  cdf.data_ptr(), width)
    if rc:
        raise HostTablesError(f"hific_build_cdf_rows: {_ERR.get(rc, rc)}")
    return cdf


def build_prior_tables(scale_table, standardized_cdf, standardized_quantile, tail_mass=2 ** (-8), precision=16):
    """prior_model.py:77-120 (conditional Gaussian / logistic prior over the latents y).
    scale_table: (n_scales,) float tensor (already lower-bounded); the two callables are the distribution's
    standardised CDF / quantile as in the reference (`maths.standardized_CDF_gaussian`, scipy ppf)."""
    scale_table = scale_table.detach().cpu()
    multiplier = -standardized_quantile(tail_mass / 2)
    pmf_center = torch.ceil(scale_table * multiplier).to(torch.int32)
    pmf_length = 2 * pmf_center + 1
    max_length = int(torch.max(pmf_length).item())
    samples = torch.abs(torch.arange(max_length).int() - pmf_center[:, None]).float()
    samples_scale = scale_table.unsqueeze(1).float()
    upper = standardized_cdf((.5 - samples) / samples_scale)
    lower = standardized_cdf((-.5 - samples) / samples_scale)
    pmf = upper - lower
    tail = (2 * lower[:, :1]).reshape(-1)
    cdf = build_cdf_rows(pmf, pmf_length, tail, precision)
    return cdf, (-pmf_center).to(torch.int32), (pmf_length + 2).to(torch.int32)


def build_hyperprior_tables(likelihood_fn, lower_tail, upper_tail, precision=16):
    """hyperprior_model.py:42-105 (factorised prior over the hyperlatents z).
    likelihood_fn(samples[C,1,L]) -> pmf [C,1,L] (the density's `likelihood(..., collapsed_format=True)`);
    lower_tail / upper_tail: (C,) tensors (`distribution.lower_tail/upper_tail(tail_mass)`)."""
    offsets = 0.
    minima = torch.clamp(torch.ceil(offsets - lower_tail.detach().cpu()).to(torch.int32), min=0)
    maxima = torch.clamp(torch.ceil(upper_tail.detach().cpu() - offsets).to(torch.int32), min=0)
    pmf_start = offsets - minima.to(torch.float32)
    pmf_length = maxima + minima + 1
    max_length = int(pmf_length.max().item())
    samples = torch.arange(max_length, dtype=torch.float32).view(1, -1) + pmf_start.view(-1, 1, 1)
    pmf = torch.squeeze(likelihood_fn(samples).detach().cpu().float())
    if pmf.dim() == 1:
        pmf = pmf.unsqueeze(0)
    # overflow mass of each row with torch's own float32 sum (vectorised pairwise reduction), as the reference does
    overflow = torch.stack([torch.clamp(1. - torch.sum(pmf[r, :int(pmf_length[r])], dim=0), min=0.)
                            for r in range(pmf.shape[0])])
    cdf = build_cdf_rows(pmf, pmf_length, overflow, precision)
    return cdf, (-minima).to(torch.int32), (pmf_length + 2).to(torch.int32)


# ---- factorised hyperlatent density on the host (table construction only; the training path runs in entropy.hip) -----
TAIL_MASS = 2 ** (-8)          # entropy_models.TAIL_MASS
MIN_LIKELIHOOD = 1e-9


def _density_params(params, prefix=""):
    """H_k, a_k, b_k (k = 0..3) as float32 CPU tensors from a module / state_dict-like mapping."""
    get = (lambda n: getattr(params, n)) if not isinstance(params, dict) else (lambda n: params[prefix + n])
    return [tuple(get(f"{n}_{k}").detach().float().cpu() for n in ("H", "a", "b")) for k in range(4)]


def _cdf_logits(par, x):
    """hyperprior_model.py:305-326; x (C,1,L)."""
    logits = x
    for H, a, b in par:
        logits = torch.bmm(torch.nn.functional.softplus(H), logits)
        logits = logits + b
        logits = logits + torch.tanh(a) * torch.tanh(logits)
    return logits


def _density_likelihood(par, x):
    """hyperprior_model.py:349-377 in collapsed (C,1,L) format."""
    up = _cdf_logits(par, x + 0.5)
    lo = _cdf_logits(par, x - 0.5)
    sign = -torch.sign(up + lo)
    lik = torch.abs(torch.sigmoid(sign * up) - torch.sigmoid(sign * lo))
    return torch.clamp(lik, min=MIN_LIKELIHOOD)                       # LowerBoundToward forward


def estimate_tails(cdf, target, shape, extra_counts=24):
    """compression_utils.py:30-80: Adam iteration (lr 1e-2, betas .9/.99) for x with cdf(x) == target, run until
    every element has passed its optimum by `extra_counts` steps.  Same float32 operation sequence as the reference."""
    lr, eps, beta_1, beta_2 = 1e-2, 1e-8, 0.9, 0.99
    tails = torch.zeros(shape, dtype=torch.float32, requires_grad=True)
    m = torch.zeros(shape, dtype=torch.float32)
    v = torch.ones(shape, dtype=torch.float32)
    counts = torch.zeros(shape, dtype=torch.int32)
    while torch.min(counts) < extra_counts:
        loss = abs(cdf(tails) - target)
        loss.backward(torch.ones_like(tails))
        tgrad = tails.grad
        with torch.no_grad():
            m = beta_1 * m + (1. - beta_1) * tgrad
            v = beta_2 * v + (1. - beta_2) * torch.square(tgrad)
            tails -= lr * m / (torch.sqrt(v) + eps)
        counts = torch.where(torch.logical_or(counts > 0, tgrad * tails.detach() > 0), counts + 1, counts)
        tails.grad.zero_()
    return tails.detach()


def build_hyperprior_tables_from_params(params, prefix="", tail_mass=TAIL_MASS, precision=16):
    """`HyperpriorEntropyModel.build_tables` (hyperprior_model.py:42-105) from the density's parameters
    (`Hyperprior.hyperlatent_likelihood` module, or a state_dict with `prefix`): tails by `estimate_tails`
    (:331-341), pmf by the density's likelihood, rows by the native quantiser."""
    import math
    par = _density_params(params, prefix)
    C = par[0][0].shape[0]
    cdf_fn = lambda x: _cdf_logits(par, x)
    lower = estimate_tails(cdf_fn, -math.log(2. / tail_mass - 1.), (C, 1, 1)).reshape(C)
    upper = estimate_tails(cdf_fn, math.log(2. / tail_mass - 1.), (C, 1, 1)).reshape(C)
    with torch.no_grad():
        return build_hyperprior_tables(lambda s: _density_likelihood(par, s), lower, upper, precision)


def host_version():
    return _load().hific_host_version().decode()

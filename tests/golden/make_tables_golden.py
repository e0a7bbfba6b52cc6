"""Golden vectors for the host table builder (tests/test_host_tables.py), made by the reference itself.
Run in the container that has /root/reference:  python tests/golden/make_tables_golden.py
Stores: random / adversarial pmfs with the reference's `maths.pmf_to_quantized_cdf` output, the reference's
prior-model tables (64 scales), and one perturbed 32-channel hyperprior density's pmf rows + tables."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_loader
ref_loader.load()
from src.helpers import maths
from src.compression import prior_model, hyperprior_model

torch.manual_seed(1234)
out = {}
pmfs, cdfs, precs = [], [], []
for trial in range(120):
    n = int(torch.randint(2, 60, (1,))); prec = int(torch.randint(8, 17, (1,))); kind = trial % 4
    if kind == 0: pmf = torch.rand(n)
    elif kind == 1: pmf = torch.softmax(torch.randn(n) * 4, 0)
    elif kind == 2:
        pmf = torch.softmax(torch.randn(n) * 6, 0); pmf[torch.rand(n) < 0.3] = 0.0
        if pmf.sum() == 0: pmf[0] = 1.
    else:
        x = torch.arange(n).float() - n / 2; pmf = torch.exp(-0.5 * (x / (0.3 + trial % 7)) ** 2); pmf = pmf / pmf.sum()
    try:
        cdf = maths.pmf_to_quantized_cdf(pmf, prec)
    except AssertionError:
        cdf = torch.full((n + 1,), -1, dtype=torch.int64)      # the reference asserts (nothing left to steal from)
    pmfs.append(pmf.numpy()); cdfs.append(cdf.numpy()); precs.append(prec)
out["n_cases"] = np.array(len(pmfs))
for i, (p, c, pr) in enumerate(zip(pmfs, cdfs, precs)):
    out[f"pmf_{i}"] = p.astype(np.float32); out[f"cdf_{i}"] = c.astype(np.int64); out[f"prec_{i}"] = np.array(pr)

pd = prior_model.PriorDensity(n_channels=220, scale_lower_bound=0.11, likelihood_type='gaussian')
pem = prior_model.PriorEntropyModel(distribution=pd, min_scale=0.11)
out["prior_scale_table"] = pem.scale_table.numpy().astype(np.float32)
out["prior_tail_mass"] = np.array(pem.tail_mass); out["prior_precision"] = np.array(pem.precision)
out["prior_CDF"] = pem.CDF.data.numpy(); out["prior_CDF_offset"] = pem.CDF_offset.data.numpy()
out["prior_CDF_length"] = pem.CDF_length.data.numpy()

hd = hyperprior_model.HyperpriorDensity(n_channels=32)
with torch.no_grad():
    for p in hd.parameters(): p.add_(0.3 * torch.randn_like(p))
hem = hyperprior_model.HyperpriorEntropyModel(distribution=hd)
hem.build_tables()
lt = hd.lower_tail(hem.tail_mass).cpu(); ut = hd.upper_tail(hem.tail_mass).cpu()
minima = torch.clamp(torch.ceil(0. - lt).to(torch.int32), min=0); maxima = torch.clamp(torch.ceil(ut - 0.).to(torch.int32), min=0)
pmf_length = maxima + minima + 1
samples = torch.arange(int(pmf_length.max()), dtype=torch.float32).view(1, -1) + (0. - minima.float()).view(-1, 1, 1)
pmf = torch.squeeze(hd.likelihood(samples, collapsed_format=True).detach().cpu())
overflow = torch.stack([torch.clamp(1. - torch.sum(pmf[r, :int(pmf_length[r])], dim=0), min=0.) for r in range(pmf.shape[0])])
out["hyper_pmf"] = pmf.numpy().astype(np.float32); out["hyper_lengths"] = pmf_length.numpy().astype(np.int32)
out["hyper_overflow"] = overflow.numpy().astype(np.float32); out["hyper_precision"] = np.array(hem.precision)
out["hyper_lower_tail"] = lt.numpy(); out["hyper_upper_tail"] = ut.numpy()
out["hyper_CDF"] = hem.CDF.data.numpy(); out["hyper_CDF_offset"] = hem.CDF_offset.data.numpy()
out["hyper_CDF_length"] = hem.CDF_length.data.numpy()
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tables_golden.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path), "bytes")

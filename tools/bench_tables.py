"""Host table construction: native port (libhific_host.so) vs the reference's Python (when /root/reference exists).
Workload: the prior-model tables of HiFIC (64 scales x up to 1479 pmf entries, precision 16) built 3 times."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, scipy.stats, torch
from hific_amd.compression import tables

g = np.load(os.path.join(ROOT, "tests", "golden", "tables_golden.npz"))
scale_table = torch.from_numpy(g["prior_scale_table"])
std_cdf = lambda x: 0.5 * torch.erfc(-(2 ** -0.5) * x)
std_q = lambda q: scipy.stats.norm.ppf(q)
def native():
    return tables.build_prior_tables(scale_table, std_cdf, std_q, float(g["prior_tail_mass"]), int(g["prior_precision"]))
native()
t = time.perf_counter()
for _ in range(3): cdf, _, _ = native()
t_native = (time.perf_counter() - t) / 3
assert np.array_equal(cdf.numpy(), g["prior_CDF"])
line = {"workload": "prior-model tables, 64 scales, precision 16", "native_ms": round(t_native * 1e3, 2)}
if os.path.isdir("/root/reference/src"):
    import ref_loader; ref_loader.load()
    from src.compression import prior_model
    import io, contextlib
    t = time.perf_counter()
    with contextlib.redirect_stderr(io.StringIO()):
        pd = prior_model.PriorDensity(n_channels=220, scale_lower_bound=0.11, likelihood_type='gaussian')
        pem = prior_model.PriorEntropyModel(distribution=pd, min_scale=0.11)
    line["reference_ms"] = round((time.perf_counter() - t) * 1e3, 1)
    line["speedup"] = round(line["reference_ms"] / line["native_ms"], 1)
print(line)

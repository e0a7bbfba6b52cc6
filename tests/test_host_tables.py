"""Host-side entropy-table construction: the C++ port in libhific_host.so must reproduce the
reference's `maths.pmf_to_quantized_cdf` and `build_tables` outputs bit for bit (they are part of the .hfc
bitstream contract).  Golden vectors were made by the reference itself (tests/golden/make_tables_golden.py); the
live comparison against the imported reference runs only where /root/reference exists."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

GOLD = os.path.join(ROOT, "tests", "golden", "tables_golden.npz")
HAVE_REF = os.path.isdir("/root/reference/src")


@pytest.fixture(scope="module")
def tables():
    from hific_amd.compression import tables as t
    return t


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def test_pmf_to_quantized_cdf_golden(tables, gold):
    n_assert = 0
    for i in range(int(gold["n_cases"])):
        pmf = torch.from_numpy(gold[f"pmf_{i}"]); want = gold[f"cdf_{i}"]; prec = int(gold[f"prec_{i}"])
        if want[0] == -1:                       # the reference raised its `best_steal != -1` assertion
            with pytest.raises(tables.HostTablesError):
                tables.pmf_to_quantized_cdf(pmf, prec)
            n_assert += 1
            continue
        got = tables.pmf_to_quantized_cdf(pmf, prec).numpy()
        assert got.dtype == np.int64 and np.array_equal(got, want), i
        assert got[0] == 0 and got[-1] == 1 << prec and np.all(np.diff(got) >= 0)
    assert n_assert < int(gold["n_cases"]) // 2


def test_argument_errors(tables):
    with pytest.raises(tables.HostTablesError):
        tables.pmf_to_quantized_cdf(torch.tensor([0.5, 0.5]), 7)            # precision < 8 (maths.py:22)
    with pytest.raises(tables.HostTablesError):
        tables.pmf_to_quantized_cdf(torch.tensor([1.0]), 16)                # fewer than 2 entries (maths.py:23)
    with pytest.raises(tables.HostTablesError):
        tables.pmf_to_quantized_cdf(torch.tensor([0.5, -0.1, 0.6]), 16)     # negative mass (maths.py:24)
    with pytest.raises(tables.HostTablesError):
        tables.pmf_to_quantized_cdf(torch.tensor([0.5, float("nan")]), 16)


def test_zero_mass_symbols_get_a_count(tables):
    pmf = torch.tensor([0.0, 0.7, 0.0, 0.3, 0.0])
    cdf = tables.pmf_to_quantized_cdf(pmf, 8).numpy()
    assert cdf[0] == 0 and cdf[-1] == 256 and np.all(np.diff(cdf) >= 1)      # every symbol codable


def test_prior_tables_golden(tables, gold):
    """prior_model.py:77-120 with the reference's 64-entry scale table and the Gaussian standardised CDF."""
    import scipy.stats
    std_cdf = lambda x: 0.5 * torch.erfc(-(2 ** -0.5) * x)                   # maths.py:87-93
    std_q = lambda q: scipy.stats.norm.ppf(q)
    cdf, off, ln = tables.build_prior_tables(torch.from_numpy(gold["prior_scale_table"]), std_cdf, std_q,
                                             tail_mass=float(gold["prior_tail_mass"]),
                                             precision=int(gold["prior_precision"]))
    assert np.array_equal(off.numpy(), gold["prior_CDF_offset"]) and np.array_equal(ln.numpy(), gold["prior_CDF_length"])
    assert cdf.dtype == torch.int32 and np.array_equal(cdf.numpy(), gold["prior_CDF"])


def test_hyperprior_rows_golden(tables, gold):
    """hyperprior_model.py:87-94 on the stored pmf rows of a perturbed 32-channel density."""
    cdf = tables.build_cdf_rows(torch.from_numpy(gold["hyper_pmf"]), torch.from_numpy(gold["hyper_lengths"]),
                                torch.from_numpy(gold["hyper_overflow"]), int(gold["hyper_precision"]))
    assert np.array_equal(cdf.numpy(), gold["hyper_CDF"])
    lt, ut = torch.from_numpy(gold["hyper_lower_tail"]), torch.from_numpy(gold["hyper_upper_tail"])
    pmf = torch.from_numpy(gold["hyper_pmf"])
    cdf2, off, ln = tables.build_hyperprior_tables(lambda s: pmf.unsqueeze(1), lt, ut, int(gold["hyper_precision"]))
    assert np.array_equal(cdf2.numpy(), gold["hyper_CDF"])
    assert np.array_equal(off.numpy(), gold["hyper_CDF_offset"]) and np.array_equal(ln.numpy(), gold["hyper_CDF_length"])


@pytest.mark.skipif(not HAVE_REF, reason="needs the reference checkout")
def test_against_imported_reference(tables):
    import ref_loader
    ref_loader.load()
    from src.helpers import maths
    torch.manual_seed(7)
    for trial in range(60):
        n = int(torch.randint(2, 50, (1,))); prec = int(torch.randint(8, 17, (1,)))
        pmf = torch.softmax(torch.randn(n) * (1 + trial % 6), 0)
        if trial % 3 == 0:
            pmf[torch.rand(n) < 0.25] = 0.0
            if pmf.sum() == 0:
                pmf[0] = 1.0
        try:
            want = maths.pmf_to_quantized_cdf(pmf, prec)
        except AssertionError:
            with pytest.raises(tables.HostTablesError):
                tables.pmf_to_quantized_cdf(pmf, prec)
            continue
        assert torch.equal(tables.pmf_to_quantized_cdf(pmf, prec), want), trial


@pytest.mark.skipif(not HAVE_REF, reason="needs the reference checkout")
def test_hyperprior_tables_from_parameters_equal_reference(tables):
    """Tail estimation (Adam iteration) + density + quantiser, all on our side, against the reference's
    `HyperpriorEntropyModel.build_tables` for the same (perturbed) density parameters."""
    import contextlib, io
    import ref_loader
    ref_loader.load()
    from src.compression import hyperprior_model
    torch.manual_seed(3)
    hd = hyperprior_model.HyperpriorDensity(n_channels=24)
    with torch.no_grad():
        for p in hd.parameters():
            p.add_(0.3 * torch.randn_like(p))
    hem = hyperprior_model.HyperpriorEntropyModel(distribution=hd)
    with contextlib.redirect_stderr(io.StringIO()):
        hem.build_tables()
    cdf, off, ln = tables.build_hyperprior_tables_from_params(hd, tail_mass=hem.tail_mass, precision=hem.precision)
    assert torch.equal(off, hem.CDF_offset.data) and torch.equal(ln, hem.CDF_length.data)
    assert torch.equal(cdf, hem.CDF.data)
    sd = {"x." + k: v for k, v in hd.state_dict().items()}
    cdf2, _, _ = tables.build_hyperprior_tables_from_params(sd, prefix="x.", tail_mass=hem.tail_mass, precision=hem.precision)
    assert torch.equal(cdf2, cdf)

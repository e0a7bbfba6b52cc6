"""Host rANS coder: the native port in libhific_host.so reproduces the reference's bitstreams
byte for byte - scalar and vectorised paths - and its decoders return what the reference's decoders return (including
the vectorised path's documented loss on multi-nibble overflows).  Goldens: tests/golden/make_rans_golden.py (the
reference's own codec under the two environment shims of oracle/ref_codec_shims.py)."""
import contextlib
import io
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
HAVE_REF = os.path.isdir("/root/reference/src")


@pytest.fixture(scope="module")
def rans():
    from hific_amd.compression import rans as r
    return r


@pytest.fixture(scope="module")
def tabs():
    g = np.load(os.path.join(ROOT, "tests", "golden", "tables_golden.npz"))
    return {n: (g[n + "_CDF"].astype(np.uint32), g[n + "_CDF_length"].astype(np.int32), g[n + "_CDF_offset"].astype(np.int32))
            for n in ("prior", "hyper")}


def test_golden_bitstreams(rans, tabs):
    g = np.load(os.path.join(ROOT, "tests", "golden", "rans_golden.npz"))
    lossy = 0
    for k in range(int(g["n_cases"])):
        cdf, cl, co = tabs[str(g[f"table_{k}"])]
        sym, idx = g[f"sym_{k}"], g[f"idx_{k}"]
        enc, shape = rans.ans_compress(sym, idx, cdf, cl, co, sym.shape[1:], 16, vectorize=False, block_encode=True)
        assert enc.dtype == np.uint32 and np.array_equal(enc, g[f"enc_scalar_{k}"]), k
        dec = rans.ans_decompress(enc, idx, cdf, cl, co, shape, 16, vectorize=False, block_decode=True)
        assert np.array_equal(dec, g[f"dec_scalar_{k}"]) and np.array_equal(dec, sym), k     # scalar path is lossless
        enc_v, cshape = rans.ans_compress(sym, idx, cdf, cl, co, sym.shape[1:], 16, vectorize=True)
        assert np.array_equal(enc_v, g[f"enc_vec_{k}"]) and tuple(cshape) == tuple(g[f"cshape_vec_{k}"]), k
        dec_v = rans.ans_decompress(enc_v, idx, cdf, cl, co, cshape, 16, vectorize=True)
        assert np.array_equal(dec_v, g[f"dec_vec_{k}"]), k
        lossy += int(not np.array_equal(dec_v, sym))
    assert 0 < lossy < int(g["n_cases"])        # the far-overflow cases decode lossily, exactly as in the reference


def test_roundtrip_properties(rans, tabs):
    rng = np.random.default_rng(5)
    cdf, cl, co = tabs["prior"]
    for shape in ((1, 220, 16, 16), (4, 16, 8, 8), (1, 1, 1, 1)):
        idx = rng.integers(0, cdf.shape[0], shape).astype(np.int32)
        width = (cl[idx] - 2).astype(np.int64)
        sym = (co[idx] + rng.integers(0, 1 << 30, shape) % np.maximum(width, 1)).astype(np.int32)   # all in range
        sym.flat[::17] = (co[idx] + width).flat[::17] + rng.integers(0, 7, sym.flat[::17].shape)   # one-nibble overflows
        for vec in (False, True):
            enc, cs = rans.ans_compress(sym, idx, cdf, cl, co, shape[1:], 16, vectorize=vec)
            dec = rans.ans_decompress(enc, idx, cdf, cl, co, cs, 16, vectorize=vec)
            assert np.array_equal(dec, sym), (shape, vec)
    sym = rng.integers(-5000, 5000, (2, 3, 4, 4)).astype(np.int32)                                  # scalar: any int32
    idx = rng.integers(0, cdf.shape[0], sym.shape).astype(np.int32)
    enc, cs = rans.ans_compress(sym, idx, cdf, cl, co, sym.shape[1:], 16, vectorize=False)
    assert np.array_equal(rans.ans_decompress(enc, idx, cdf, cl, co, cs, 16, vectorize=False), sym)
    per = rans.ans_compress(sym, idx, cdf, cl, co, sym.shape[1:], 16, vectorize=False, block_encode=False)
    assert len(per) == 2
    assert np.array_equal(rans.ans_decompress(per, idx, cdf, cl, co, cs, 16, vectorize=False, block_decode=False), sym)


def test_errors_are_loud(rans, tabs):
    cdf, cl, co = tabs["hyper"]
    sym = np.zeros((1, 2, 2, 2), np.int32); idx = np.zeros_like(sym)
    bad = idx.copy(); bad.flat[3] = cdf.shape[0]
    with pytest.raises(rans.RansError):
        rans.ans_compress(sym, bad, cdf, cl, co, (2, 2, 2), 16)
    with pytest.raises(rans.RansError):
        rans.ans_compress(sym, idx, cdf, cl, co, (2, 2, 2), 40)
    enc, cs = rans.ans_compress(sym + 100000, idx, cdf, cl, co, (2, 2, 2), 16)
    with pytest.raises(rans.RansError):
        rans.ans_decompress(enc[:3], idx, cdf, cl, co, cs, 16)           # truncated message
    with pytest.raises(rans.RansError):
        rans.ans_compress(sym[0], idx[0], cdf, cl, co, (2, 2, 2), 16)    # not (N,C,H,W)


@pytest.mark.skipif(not HAVE_REF, reason="needs the reference checkout")
def test_against_reference_codec(rans, tabs):
    import ref_loader, ref_codec_shims
    ref_loader.load()
    _, cu = ref_codec_shims.apply()
    rng = np.random.default_rng(99)
    for name, (cdf, cl, co) in tabs.items():
        for shape in ((1, 6, 3, 4), (2, 5, 2, 3)):
            idx = rng.integers(0, cdf.shape[0], shape).astype(np.int32)
            sym = np.round(rng.normal(0, 6, shape)).astype(np.int32)
            sym.flat[0] = 70000; sym.flat[5] = -12345
            for vec in (False, True):
                with contextlib.redirect_stdout(io.StringIO()):
                    enc_r, cs_r = cu.ans_compress(sym, idx, cdf, cl, co, shape[1:], precision=16, vectorize=vec, block_encode=True)
                    dec_r = cu.ans_decompress(enc_r, idx, cdf, cl, co, cs_r, precision=16, vectorize=vec, block_decode=True)
                enc, cs = rans.ans_compress(sym, idx, cdf, cl, co, shape[1:], 16, vectorize=vec)
                assert np.array_equal(enc, np.asarray(enc_r, dtype=np.uint32)) and tuple(cs) == tuple(cs_r)
                dec = rans.ans_decompress(enc, idx, cdf, cl, co, cs, 16, vectorize=vec)
                assert np.array_equal(dec, np.asarray(dec_r).reshape(shape).astype(np.int32))


def _container_case():
    rng = np.random.default_rng(3)
    from hific_amd.compression import container
    co = container.CompressionOutput(
        hyperlatents_encoded=rng.integers(0, 2 ** 32, 321, dtype=np.uint64).astype(np.uint32),
        latents_encoded=rng.integers(0, 2 ** 32, 4567, dtype=np.uint64).astype(np.uint32),
        hyperlatent_spatial_shape=(4, 6), batch_shape=1, spatial_shape=(256, 384),
        hyper_coding_shape=(320, 1, 1), latent_coding_shape=(220, 1, 1))
    return container, co


def test_hfc_container_roundtrip_and_layout(tmp_path):
    container, co = _container_case()
    path = str(tmp_path / "a.hfc")
    actual_bpp, _ = container.save_compressed_format(co, path)
    raw = open(path, "rb").read()
    assert len(raw) == 2 * 11 + 4 + (4 + 4 * 321 + 4) + (4 + 4 * 4567 + 4)
    assert raw[22:26] == b"\x46\xE2\x84\x92" and raw[-4:] == b"\x46\xE2\x84\x92"
    assert abs(actual_bpp - 8.0 * len(raw) / (256 * 384)) < 1e-12
    back = container.load_compressed_format(path)
    assert back.hyperlatent_spatial_shape == (4, 6) and back.spatial_shape == (256, 384) and back.batch_shape == 1
    assert back.hyper_coding_shape == (320, 1, 1) and back.latent_coding_shape == (220, 1, 1)
    assert np.array_equal(back.hyperlatents_encoded, co.hyperlatents_encoded)
    assert np.array_equal(back.latents_encoded, co.latents_encoded)
    open(path, "wb").write(raw[:100])
    with pytest.raises(container.ContainerError):
        container.load_compressed_format(path)


@pytest.mark.skipif(not HAVE_REF, reason="needs the reference checkout")
def test_hfc_container_bytes_equal_reference(tmp_path):
    import ref_loader, ref_codec_shims
    ref_loader.load()
    _, cu = ref_codec_shims.apply()
    container, co = _container_case()
    ours, theirs = str(tmp_path / "ours.hfc"), str(tmp_path / "ref.hfc")
    container.save_compressed_format(co, ours)
    from types import SimpleNamespace
    cu.save_compressed_format(SimpleNamespace(total_bpp=0.5, **co._asdict()), theirs)
    assert open(ours, "rb").read() == open(theirs, "rb").read()
    back = cu.load_compressed_format(ours)                      # the reference reads what we write
    assert np.array_equal(back.latents_encoded, co.latents_encoded) and back.latent_coding_shape == (220, 1, 1)


def test_vectorised_coder_accepts_torch_tensors_and_returns_on_device(rans, tabs):
    """The vectorised coder takes torch tensors (the device symbol kernels' outputs) and, with `device=...`, returns a tensor:
    same bitstream and symbols as the numpy path, for both step layouts (batch 1: steps = pixels; batch > 1: steps = images).
    (The CUDA flavour of the same check is tests/test_gpu_elementwise.py::test_vectorised_coder_takes_device_tensors.)"""
    import torch
    cdf, cl, co = tabs["prior"]
    for shape in ((1, 24, 6, 5), (3, 8, 4, 4)):
        gen = torch.Generator().manual_seed(3)
        idx = torch.randint(0, cdf.shape[0], shape, generator=gen, dtype=torch.int32)
        sym = torch.round(torch.randn(shape, generator=gen) * 2).to(torch.int32)
        enc_n, cs_n = rans.ans_compress(sym.numpy(), idx.numpy(), cdf, cl, co, shape[1:], 16, vectorize=True)
        enc_t, cs_t = rans.ans_compress(sym, idx, cdf, cl, co, shape[1:], 16, vectorize=True)
        assert tuple(cs_n) == tuple(cs_t) and np.array_equal(enc_n, enc_t)
        a, _, _ = rans._steps_layout(sym)
        b, _, _ = rans._steps_layout(sym.numpy())
        assert np.array_equal(a, b)
        dec_n = rans.ans_decompress(enc_n, idx.numpy(), cdf, cl, co, cs_n, 16, vectorize=True)
        dec_t = rans.ans_decompress(enc_n, idx, cdf, cl, co, cs_n, 16, vectorize=True, device="cpu")
        assert isinstance(dec_t, torch.Tensor) and tuple(dec_t.shape) == shape
        assert np.array_equal(dec_t.numpy(), dec_n) and np.array_equal(dec_n, sym.numpy())


@pytest.mark.parametrize("precision", [8, 12, 16, 20, 24])
def test_synthetic_tables_all_precisions_and_both_symbol_searches(rans, precision):
    """Random cdf tables at every supported precision: the scalar coder is lossless for any int32 symbol, the vectorised one
    for in-range symbols; a long message (table-guided symbol search in the decoders) and a short one (binary search) decode
    to the same symbols; > 16 bits of precision takes the divide instruction instead of the reciprocal table."""
    rng = np.random.default_rng(precision)
    rows, maxlen = 12, min(90, (1 << precision) // 4)
    stride = maxlen + 2
    cdf = np.zeros((rows, stride), np.uint32); cl = np.zeros(rows, np.int32)
    co = rng.integers(-9, 3, rows).astype(np.int32)
    for r in range(rows):
        n = int(rng.integers(3, maxlen + 1))
        w = rng.integers(1, 40, n).astype(np.float64)
        f = np.maximum(1, np.floor(w / w.sum() * (1 << precision))).astype(np.int64)
        f[np.argmax(f)] += (1 << precision) - f.sum()
        cdf[r, :n + 1] = np.concatenate([[0], np.cumsum(f)]); cl[r] = n + 1
    for shape in ((1, 7, 3, 4), (1, 40, 12, 12), (2, 9, 5, 5)):       # 84 / 5760 / 450 symbols
        idx = rng.integers(0, rows, shape).astype(np.int32)
        wild = np.round(rng.normal(0, 300, shape)).astype(np.int32)               # mostly out of range: overflow codes
        enc, cs = rans.ans_compress(wild, idx, cdf, cl, co, shape[1:], precision, vectorize=False)
        assert np.array_equal(rans.ans_decompress(enc, idx, cdf, cl, co, cs, precision, vectorize=False), wild)
        inr = (co[idx] + rng.integers(0, 1 << 30, shape) % np.maximum(cl[idx] - 2, 1)).astype(np.int32)   # inside the table
        for vec in (False, True):
            enc, cs = rans.ans_compress(inr, idx, cdf, cl, co, shape[1:], precision, vectorize=vec)
            assert np.array_equal(rans.ans_decompress(enc, idx, cdf, cl, co, cs, precision, vectorize=vec), inr), (shape, vec)

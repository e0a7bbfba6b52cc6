"""Golden bitstreams for the native rANS coder (tests/test_host_rans.py), made by the reference's own codec
(with the two environment shims of oracle/ref_codec_shims.py).  Run where /root/reference exists:
    python tests/golden/make_rans_golden.py"""
import contextlib, io, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_loader, ref_codec_shims
ref_loader.load()
_, cu = ref_codec_shims.apply()

g = np.load(os.path.join(ROOT, "tests", "golden", "tables_golden.npz"))
rng = np.random.default_rng(2024)
out = {}
k = 0
for name in ("prior", "hyper"):
    cdf = g[name + "_CDF"].astype(np.uint32); cl = g[name + "_CDF_length"].astype(np.int32); co = g[name + "_CDF_offset"].astype(np.int32)
    for trial in range(8):
        shape = [(1, 4, 3, 5), (2, 3, 4, 4), (1, 8, 2, 2), (3, 2, 1, 7)][trial % 4]
        idx = rng.integers(0, cdf.shape[0], shape).astype(np.int32)
        sym = np.round(rng.normal(0, 1 + 2 * trial, shape)).astype(np.int32)
        if trial % 2 == 0:      # far overflows (multi-nibble), near overflows, exact max_value
            sym.flat[0] = 2000 + trial; sym.flat[-1] = -3000 - 7 * trial
            m = sym.size // 2; sym.flat[m] = cl[idx.flat[m]] - 2 + co[idx.flat[m]]
            sym.flat[1] = cl[idx.flat[1]] - 2 + co[idx.flat[1]] + 3; sym.flat[2] = co[idx.flat[2]] - 2
        with contextlib.redirect_stdout(io.StringIO()):
            enc_s, _ = cu.ans_compress(sym, idx, cdf, cl, co, shape[1:], precision=16, vectorize=False, block_encode=True)
            dec_s = cu.ans_decompress(enc_s, idx, cdf, cl, co, shape[1:], precision=16, vectorize=False, block_decode=True)
            enc_v, cshape = cu.ans_compress(sym, idx, cdf, cl, co, shape[1:], precision=16, vectorize=True, block_encode=True)
            dec_v = cu.ans_decompress(enc_v, idx, cdf, cl, co, cshape, precision=16, vectorize=True, block_decode=True)
        out[f"table_{k}"] = np.array(name); out[f"sym_{k}"] = sym; out[f"idx_{k}"] = idx
        out[f"enc_scalar_{k}"] = np.asarray(enc_s, dtype=np.uint32)
        out[f"dec_scalar_{k}"] = np.asarray(dec_s).reshape(shape).astype(np.int32)
        out[f"enc_vec_{k}"] = np.asarray(enc_v, dtype=np.uint32); out[f"dec_vec_{k}"] = np.asarray(dec_v).astype(np.int32)
        out[f"cshape_vec_{k}"] = np.array(cshape)
        k += 1
out["n_cases"] = np.array(k)
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rans_golden.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path), "bytes,", k, "cases")

"""Host rANS coder: native port (libhific_host.so) vs the reference's Python codec (where /root/reference exists,
under the shims of oracle/ref_codec_shims.py).  Workload: one 256x256 image's latents (1,220,16,16) against the
64-row prior tables and its hyperlatents (1,320,4,4) against a 32-row hyperprior table, precision 16."""
import contextlib, io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from hific_amd.compression import rans

g = np.load(os.path.join(ROOT, "tests", "golden", "tables_golden.npz"))
rng = np.random.default_rng(0)
def workload(name, shape, sigma):
    cdf = g[name + "_CDF"].astype(np.uint32); cl = g[name + "_CDF_length"].astype(np.int32); co = g[name + "_CDF_offset"].astype(np.int32)
    idx = rng.integers(0, cdf.shape[0], shape).astype(np.int32)
    sym = np.round(rng.normal(0, sigma, shape)).astype(np.int32)
    return sym, idx, cdf, cl, co
cu = None
if os.path.isdir("/root/reference/src"):
    import ref_loader, ref_codec_shims
    ref_loader.load(); _, cu = ref_codec_shims.apply()
for name, shape, sigma in (("prior", (1, 220, 16, 16), 3.0), ("hyper", (1, 320, 4, 4), 2.0)):
    sym, idx, cdf, cl, co = workload(name, shape, sigma)
    for vec in (False, True):
        reps = 20
        t = time.perf_counter()
        for _ in range(reps): enc, cs = rans.ans_compress(sym, idx, cdf, cl, co, shape[1:], 16, vectorize=vec)
        te = (time.perf_counter() - t) / reps
        t = time.perf_counter()
        for _ in range(reps): dec = rans.ans_decompress(enc, idx, cdf, cl, co, cs, 16, vectorize=vec)
        td = (time.perf_counter() - t) / reps
        line = {"tensor": f"{name} {shape}", "vectorize": vec, "symbols": sym.size, "bytes": 4 * enc.size,
                "native_encode_ms": round(te * 1e3, 3), "native_decode_ms": round(td * 1e3, 3),
                "native_Msym_per_s": round(sym.size / te / 1e6, 1)}
        if cu is not None:
            with contextlib.redirect_stdout(io.StringIO()):
                t = time.perf_counter(); enc_r, cs_r = cu.ans_compress(sym, idx, cdf, cl, co, shape[1:], precision=16, vectorize=vec, block_encode=True); tre = time.perf_counter() - t
                t = time.perf_counter(); cu.ans_decompress(enc_r, idx, cdf, cl, co, cs_r, precision=16, vectorize=vec, block_decode=True); trd = time.perf_counter() - t
            assert np.array_equal(np.asarray(enc_r, dtype=np.uint32), enc)
            line.update({"reference_encode_ms": round(tre * 1e3, 1), "reference_decode_ms": round(trd * 1e3, 1),
                         "speedup_encode": round(tre / te), "speedup_decode": round(trd / td)})
        print(line)

"""EVALUATION path orchestration (`hific_amd.compression.codec`): compress -> .hfc -> decompress with the oracle's
CPU networks; pinned byte for byte against the reference's `Hyperprior.compress_forward` where the checkout exists."""
import contextlib
import io
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from oracle import hific_oracle as O

HAVE_REF = os.path.isdir("/root/reference/src")


def _nets(sd):
    from hific_amd.compression import codec
    return codec.CodecNets(
        analysis=lambda y: O.hyper_analysis_forward(sd, y),
        synthesis_mu=lambda z: O.hyper_synthesis_forward(sd, z, "Hyperprior.synthesis_mu."),
        synthesis_std=lambda z: O.hyper_synthesis_forward(sd, z, "Hyperprior.synthesis_std."))


def _symfns():
    from hific_amd.compression import codec
    return codec.SymbolFns(
        hyper=O.hyper_symbols_and_indices,
        prior=lambda y, m, s, tab: (O.prior_symbols(y, m), O.prior_compute_indices(s, tab)),
        prior_indices=O.prior_compute_indices)


def test_compress_container_decompress_roundtrip(tmp_path):
    from hific_amd.compression import codec, container
    g = np.load(os.path.join(ROOT, "tests", "golden", "tables_golden.npz"))
    hyper_t = codec.EntropyTables(g["hyper_CDF"], g["hyper_CDF_offset"], g["hyper_CDF_length"])      # 32 channels
    prior_t = codec.EntropyTables(g["prior_CDF"], g["prior_CDF_offset"], g["prior_CDF_length"])
    sd = {k: v for k, v in O.make_state_dict(seed=4, C=24, N=32, n_res=0, gan=False).items() if k.startswith("Hyperprior.")}
    nets, fns, tab = _nets(sd), _symfns(), O.prior_scale_table()
    for batch, vec in ((1, True), (2, True), (2, False)):
        y = O.make_noise(21 + batch, (batch, 24, 8, 12)) * 3
        out = codec.compress_forward(y, (128, 192), nets, hyper_t, prior_t, tab, fns, vectorize=vec)
        path = str(tmp_path / f"x{batch}{vec}.hfc")
        container.save_compressed_format(out, path)
        back = container.load_compressed_format(path)
        y_hat = codec.decompress_forward(back, nets, hyper_t, prior_t, tab, fns, n_hyper_channels=32, vectorize=vec)
        # what the decoder must reproduce: the latents quantised around the means predicted from the DECODED hyperlatents
        z_hat = torch.floor(nets.analysis(y) + 0.5)
        mu = nets.synthesis_mu(z_hat)
        assert y_hat.shape == y.shape and torch.equal(y_hat, torch.floor(y + 0.5 - mu) + mu)
        assert back.latent_coding_shape == ((24, 1, 1) if (vec and batch == 1) else (24, 8, 12))


@pytest.mark.skipif(not HAVE_REF, reason="needs the reference checkout")
@pytest.mark.filterwarnings("ignore::DeprecationWarning")
def test_compress_forward_bytes_equal_reference():
    """Same weights, same latents: our orchestration (oracle networks + native tables' consumers + native coder) emits
    the reference's `hyperlatents_encoded` / `latents_encoded` words exactly, for the default vectorised coder."""
    import ref_loader, ref_codec_shims
    ns = ref_loader.load()
    ref_codec_shims.apply()
    from hific_amd.compression import codec
    torch.manual_seed(0)
    C, N = 12, 16
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        hp = ns.hyperprior.Hyperprior(bottleneck_capacity=C, hyperlatent_filters=N, entropy_code=True)
        hp.hyperprior_entropy_model.build_tables()             # compress.py:61 does this after loading a checkpoint
    hp.eval()
    sd = {"Hyperprior." + k: v for k, v in hp.state_dict().items()}
    y = O.make_noise(5, (1, C, 8, 8)) * 4
    with contextlib.redirect_stdout(io.StringIO()), torch.no_grad():
        ref_out = hp.compress_forward(y, (128, 128))
    hem, pem = hp.hyperprior_entropy_model, hp.prior_entropy_model
    hyper_t = codec.EntropyTables(hem.CDF.data, hem.CDF_offset.data, hem.CDF_length.data)
    prior_t = codec.EntropyTables(pem.CDF.data, pem.CDF_offset.data, pem.CDF_length.data)
    with torch.no_grad():
        out = codec.compress_forward(y, (128, 128), _nets(sd), hyper_t, prior_t, pem.scale_table, _symfns(),
                                     vectorize=True, block_encode=True, precision=pem.precision)
        assert np.array_equal(out.hyperlatents_encoded, np.asarray(ref_out.hyperlatents_encoded, dtype=np.uint32))
        assert np.array_equal(out.latents_encoded, np.asarray(ref_out.latents_encoded, dtype=np.uint32))
        assert tuple(out.hyper_coding_shape) == tuple(ref_out.hyper_coding_shape)
        assert tuple(out.latent_coding_shape) == tuple(ref_out.latent_coding_shape)
        with contextlib.redirect_stdout(io.StringIO()):
            ref_dec = hp.decompress_forward(ref_out, device="cpu")
        ours = codec.decompress_forward(out, _nets(sd), hyper_t, prior_t, pem.scale_table, _symfns(), n_hyper_channels=N)
        assert torch.equal(ours, ref_dec)


REF_FIELDS = ("hyperlatents_encoded", "latents_encoded", "hyperlatent_spatial_shape", "batch_shape", "spatial_shape",
              "hyper_coding_shape", "latent_coding_shape", "hyperlatent_bits", "latent_bits", "total_bits",
              "hyperlatent_bpp", "latent_bpp", "total_bpp")          # src/hyperprior.py:25-39


def _oracle_bits(sd):
    """`_estimate_compression_bits` of both entropy models with the oracle's likelihoods."""
    def fn(z, y, means, scales):
        qz = torch.floor(z + 0.5)
        hb = torch.sum(torch.log(O.factorized_likelihood(sd, qz) + 1e-9)) / -np.log(2.)
        qy = torch.floor(y - means + 0.5) + means
        lb = torch.sum(torch.log(O.latent_likelihood(qy, means, scales) + 1e-9)) / -np.log(2.)
        return hb, lb
    return fn


def test_compression_output_carries_the_reference_reporting_fields(tmp_path):
    """The reference's drivers read `.total_bpp` & co. off `compress_forward`'s result (model.py:296-309,
    compress.py:92, compression_utils.save_compressed_format): all 13 fields exist, with and without a bits_fn."""
    from hific_amd.compression import codec, container
    assert codec.CompressionOutput._fields == REF_FIELDS
    g = np.load(os.path.join(ROOT, "tests", "golden", "tables_golden.npz"))
    hyper_t = codec.EntropyTables(g["hyper_CDF"], g["hyper_CDF_offset"], g["hyper_CDF_length"])
    prior_t = codec.EntropyTables(g["prior_CDF"], g["prior_CDF_offset"], g["prior_CDF_length"])
    sd = {k: v for k, v in O.make_state_dict(seed=4, C=24, N=32, n_res=0, gan=False).items() if k.startswith("Hyperprior.")}
    y = O.make_noise(22, (1, 24, 8, 12)) * 3
    with torch.no_grad():
        a = codec.compress_forward(y, (128, 192), _nets(sd), hyper_t, prior_t, O.prior_scale_table(), _symfns())
        b = codec.compress_forward(y, (128, 192), _nets(sd), hyper_t, prior_t, O.prior_scale_table(), _symfns(),
                                   bits_fn=_oracle_bits(sd))
    assert a.total_bits == 32.0 * (len(a.hyperlatents_encoded) + len(a.latents_encoded))       # attained size
    for o in (a, b):
        assert o.total_bits == o.hyperlatent_bits + o.latent_bits
        assert abs(o.total_bpp - o.total_bits / (128 * 192)) < 1e-9 and o.total_bpp > 0
        assert abs(o.total_bpp - (o.hyperlatent_bpp + o.latent_bpp)) < 1e-9
    # the estimate and the attained size describe the same message: same order of magnitude
    assert 0.3 < b.total_bits / a.total_bits < 3.0
    container.save_compressed_format(b, str(tmp_path / "o.hfc"))


@pytest.mark.skipif(not HAVE_REF, reason="needs the reference checkout")
@pytest.mark.filterwarnings("ignore::DeprecationWarning")
def test_reference_drivers_accept_our_compression_output(tmp_path):
    """The reference's own `save_compressed_format` (compression_utils.py:300-335) and its reporting consume our
    output; the Shannon estimates equal the reference's `_estimate_compression_bits`."""
    import ref_loader, ref_codec_shims
    ns = ref_loader.load()
    ref_codec_shims.apply()
    import importlib
    cu = importlib.import_module("src.compression.compression_utils")
    from hific_amd.compression import codec, container
    assert codec.CompressionOutput._fields == ns.hyperprior.CompressionOutput._fields
    torch.manual_seed(0)
    C, N = 12, 16
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        hp = ns.hyperprior.Hyperprior(bottleneck_capacity=C, hyperlatent_filters=N, entropy_code=True)
        hp.hyperprior_entropy_model.build_tables()
    hp.eval()
    sd = {"Hyperprior." + k: v for k, v in hp.state_dict().items()}
    y = O.make_noise(5, (1, C, 8, 8)) * 4
    with contextlib.redirect_stdout(io.StringIO()), torch.no_grad():
        ref_out = hp.compress_forward(y, (128, 128))
    hem, pem = hp.hyperprior_entropy_model, hp.prior_entropy_model
    hyper_t = codec.EntropyTables(hem.CDF.data, hem.CDF_offset.data, hem.CDF_length.data)
    prior_t = codec.EntropyTables(pem.CDF.data, pem.CDF_offset.data, pem.CDF_length.data)
    with torch.no_grad():
        out = codec.compress_forward(y, (128, 128), _nets(sd), hyper_t, prior_t, pem.scale_table, _symfns(),
                                     precision=pem.precision, bits_fn=_oracle_bits(sd))
    for f in ("hyperlatent_bits", "latent_bits", "total_bits", "hyperlatent_bpp", "latent_bpp", "total_bpp"):
        a, b = getattr(out, f), getattr(ref_out, f)
        assert abs(a - b) <= 1e-4 * abs(b) + 1e-6, (f, a, b)
    ref_path, our_path = str(tmp_path / "ref.hfc"), str(tmp_path / "our.hfc")
    with contextlib.redirect_stdout(io.StringIO()):
        cu.save_compressed_format(out, ref_path)            # the reference's writer on OUR namedtuple
    container.save_compressed_format(out, our_path)
    assert open(ref_path, "rb").read() == open(our_path, "rb").read()


def test_module_build_tables_matches_golden_prior_tables(hific):
    """`hific_amd.Hyperprior.build_tables()` runs on the host: its prior tables are the reference's (golden), its
    hyperprior tables are well-formed for the module's own density parameters."""
    import hific_amd
    g = np.load(os.path.join(ROOT, "tests", "golden", "tables_golden.npz"))
    torch.manual_seed(1)
    hp = hific_amd.hyperprior.Hyperprior(bottleneck_capacity=12, hyperlatent_filters=16)
    hyp, prior, scale_table = hp.build_tables()
    assert np.array_equal(prior.CDF.numpy(), g["prior_CDF"]) and np.array_equal(prior.CDF_offset.numpy(), g["prior_CDF_offset"])
    assert np.allclose(scale_table.numpy(), g["prior_scale_table"])
    assert hyp.CDF.shape[0] == 16 and hyp.CDF.dtype == torch.int32
    for r in range(16):
        row = hyp.CDF[r, :int(hyp.CDF_length[r])].numpy()
        assert row[0] == 0 and row[-1] == 1 << 16 and np.all(np.diff(row) > 0)


@pytest.mark.skipif(not HAVE_REF, reason="needs the reference checkout")
def test_module_build_tables_equal_reference(hific):
    import ref_loader
    import hific_amd
    ns = ref_loader.load()
    torch.manual_seed(2)
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        ref = ns.hyperprior.Hyperprior(bottleneck_capacity=12, hyperlatent_filters=16, entropy_code=True)
        with torch.no_grad():
            for p_ in ref.hyperlatent_likelihood.parameters():
                p_.add_(0.2 * torch.randn_like(p_))
        ref.hyperprior_entropy_model.build_tables()
    ours = hific_amd.hyperprior.Hyperprior(bottleneck_capacity=12, hyperlatent_filters=16)
    ours.load_state_dict({k: v for k, v in ref.state_dict().items()
                          if not any(t in k for t in ("CDF", "scale_table", "min_scale", "entropy_model", "prior_density"))},
                         strict=True)
    hyp, prior, _ = ours.build_tables()
    hem, pem = ref.hyperprior_entropy_model, ref.prior_entropy_model
    assert torch.equal(hyp.CDF, hem.CDF.data) and torch.equal(hyp.CDF_offset, hem.CDF_offset.data)
    assert torch.equal(prior.CDF, pem.CDF.data) and torch.equal(prior.CDF_length, pem.CDF_length.data)

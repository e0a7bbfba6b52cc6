"""Drop-in for the reference's src/hyperprior.py training/validation path (CodingModel, Hyperprior.forward,
HyperInfo): same constructor signature, attribute names (`analysis_net`, `synthesis_mu`, `synthesis_std`,
`hyperlatent_likelihood`, `amortization_models`) and the same order of noise draws from torch's global RNG
(hyperlatent noise, then latent noise: src/hyperprior.py:283,305).  All arithmetic runs in csrc/entropy.hip.

The EVALUATION-mode methods (`build_tables`, `compress_forward`, `decompress_forward`, src/hyperprior.py:183-274) are
the device modules plugged into `compression.codec` (symbols/indices from csrc/entropy.hip, tables + rANS coder from
libhific_host.so); the orchestration and the host pieces are pinned byte for byte on the CPU
(tests/test_host_codec.py), this GPU wiring has not been run on a GPU yet."""
from collections import namedtuple

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .network import hyper
from .compression import hyperprior_model

MIN_SCALE = 0.11
LOG_SCALES_MIN = -3.
MIN_LIKELIHOOD = 1e-9
MAX_LIKELIHOOD = 1e3
SMALL_HYPERLATENT_FILTERS = 192
LARGE_HYPERLATENT_FILTERS = 320

HyperInfo = namedtuple(
    "HyperInfo",
    "decoded "
    "latent_nbpp hyperlatent_nbpp total_nbpp latent_qbpp hyperlatent_qbpp total_qbpp",
)


def lower_bound_toward(x, bound):
    return ops.LowerBoundFn.apply(x.contiguous(), bound)


class CodingModel(nn.Module):
    """Entropy estimation pieces of src/hyperprior.py:44-139."""

    def __init__(self, n_channels, min_likelihood=MIN_LIKELIHOOD, max_likelihood=MAX_LIKELIHOOD):
        super().__init__()
        self.n_channels = n_channels
        self.min_likelihood = float(min_likelihood)
        self.max_likelihood = float(max_likelihood)
        self.likelihood_logistic = 0

    def _draw_noise(self, x):
        # same RNG call as the reference (_quantize: torch.nn.init.uniform_(torch.zeros_like(x), -0.5, 0.5)): uniform_ overwrites
        # every element, so the zero fill is skipped (one launch per draw); same generator consumption, same values
        with torch.no_grad():
            return torch.empty_like(x).uniform_(-0.5, 0.5)

    def _quantize(self, x, mode='noise', means=None):
        if mode == 'noise':
            return ops.AddNoiseFn.apply(x.contiguous(), self._draw_noise(x))
        if mode == 'quantize':
            return ops.RoundFn.apply(x.contiguous(), None if means is None else means.contiguous())
        raise NotImplementedError

    def _estimate_entropy(self, likelihood, spatial_shape, want_bits=True):
        """src/hyperprior.py:80-93.  `want_bits=False` (the training forward, which only uses the rate per pixel): 1 / n_pixels
        is folded into the reduction kernel's multiplier - no separate zero-dimensional division (and its backward) per
        entropy term; n_bits is then None."""
        EPS = 1e-9
        quotient = -np.log(2.)
        batch_size = likelihood.size()[0]
        assert len(spatial_shape) == 2, 'Mispecified spatial dims'
        n_pixels = np.prod(spatial_shape)
        if not want_bits:
            return None, ops.LogSumFn.apply(likelihood.contiguous(), EPS, 1.0 / (batch_size * quotient * float(n_pixels)))
        n_bits = ops.LogSumFn.apply(likelihood.contiguous(), EPS, 1.0 / (batch_size * quotient))
        bpp = n_bits / n_pixels
        return n_bits, bpp

    def quantize_latents_st(self, inputs, means=None):
        return ops.RoundSTFn.apply(inputs.contiguous(), None if means is None else means.contiguous())

    def latent_likelihood(self, x, mean, scale):
        return ops.GaussLikFn.apply(x.contiguous(), mean.contiguous(), scale.contiguous(), self.min_likelihood,
                                    self.likelihood_logistic)


class _EntropyModelHandle:
    """What the reference's driver touches on `Hyperprior.hyperprior_entropy_model` (compress.py:61,122:
    `.build_tables()` after a checkpoint load): forwards to the owning module's host-side table construction."""

    def __init__(self, owner):
        self._owner = owner

    def build_tables(self, **kwargs):
        return self._owner.build_tables()


class Hyperprior(CodingModel):
    def __init__(self, bottleneck_capacity=220, hyperlatent_filters=LARGE_HYPERLATENT_FILTERS, mode='large',
                 likelihood_type='gaussian', scale_lower_bound=MIN_SCALE, entropy_code=False,
                 vectorize_encoding=True, block_encode=True, lazy_tables=False):
        super().__init__(n_channels=bottleneck_capacity)
        self.bottleneck_capacity = bottleneck_capacity
        self.scale_lower_bound = scale_lower_bound
        if mode == 'small':
            hyperlatent_filters = SMALL_HYPERLATENT_FILTERS
        self.analysis_net = hyper.HyperpriorAnalysis(C=bottleneck_capacity, N=hyperlatent_filters)
        self.synthesis_mu = hyper.HyperpriorSynthesis(C=bottleneck_capacity, N=hyperlatent_filters)
        self.synthesis_std = hyper.HyperpriorSynthesis(C=bottleneck_capacity, N=hyperlatent_filters)
        self.amortization_models = [self.analysis_net, self.synthesis_mu, self.synthesis_std]
        # y - mu is floored into the indices: both nets that produce mu run the exact-index forward (ops.set_exact_index)
        hyper.mark_exact_index_chain(self.analysis_net)
        hyper.mark_exact_index_chain(self.synthesis_mu)
        self.hyperlatent_likelihood = hyperprior_model.HyperpriorDensity(n_channels=hyperlatent_filters)
        if likelihood_type == 'gaussian':
            self.likelihood_logistic = 0
        elif likelihood_type == 'logistic':
            self.likelihood_logistic = 1
        else:
            raise ValueError('Unknown likelihood model: {}'.format(likelihood_type))
        self.hyperlatent_filters = hyperlatent_filters
        self.likelihood_type = likelihood_type
        self.vectorize_encoding = vectorize_encoding
        self.block_encode = block_encode
        self._tables = None
        if entropy_code is True:
            object.__setattr__(self, 'hyperprior_entropy_model', _EntropyModelHandle(self))   # not a sub-module
            if not lazy_tables:          # lazy: built on the first compress/decompress (a forward never needs them)
                self.build_tables()

    # ---- EVALUATION path (src/hyperprior.py:183-274) -----------------------------------------------------------
    def build_tables(self):
        """Prior tables (64 scales, prior_model.py:77-120) and hyperprior tables from the current density parameters
        (hyperprior_model.py:42-105; the reference rebuilds these after loading a checkpoint, compress.py:61)."""
        import scipy.stats
        from .compression import codec, tables
        scale_table = torch.Tensor(np.exp(np.linspace(np.log(0.11), np.log(256), 64)))         # prior_model.py:23-25
        scale_table = torch.clamp(scale_table, min=self.scale_lower_bound)                       # :60
        if self.likelihood_logistic:
            std_cdf, std_q = torch.sigmoid, (lambda q: scipy.stats.logistic.ppf(q))
        else:
            std_cdf, std_q = (lambda x: 0.5 * torch.erfc(-(2 ** -0.5) * x)), (lambda q: scipy.stats.norm.ppf(q))
        prior = codec.EntropyTables(*tables.build_prior_tables(scale_table, std_cdf, std_q))
        hyp = codec.EntropyTables(*tables.build_hyperprior_tables_from_params(self.hyperlatent_likelihood))
        self._tables = (hyp, prior, scale_table)
        return self._tables

    def _codec_parts(self):
        from .compression import codec
        if self._tables is None:
            self.build_tables()
        nets = codec.CodecNets(self.analysis_net, self.synthesis_mu, self.synthesis_std)
        fns = codec.SymbolFns(
            hyper=ops.hyper_symbols_and_indices, prior=ops.prior_symbols_and_indices,
            prior_indices=lambda scales, tab: ops.prior_symbols_and_indices(scales, scales, scales, tab)[1])
        return nets, fns

    def _shannon_bits(self, hyperlatents, latents, means, scales):
        """`_estimate_compression_bits` of both entropy models (hyperprior_model.py:108-131, prior_model.py:122-145):
        -sum log2(likelihood(quantised) + 1e-9), on the likelihood kernels of the training path."""
        inv_q = 1.0 / -np.log(2.)
        qz = ops.RoundFn.apply(hyperlatents.contiguous(), None)
        hyper_bits = ops.LogSumFn.apply(self.hyperlatent_likelihood(qz).contiguous(), 1e-9, inv_q)
        qy = ops.RoundFn.apply(latents.contiguous(), means.contiguous())
        latent_bits = ops.LogSumFn.apply(self.latent_likelihood(qy, mean=means, scale=scales).contiguous(), 1e-9, inv_q)
        return hyper_bits.item(), latent_bits.item()

    def compress_forward(self, latents, spatial_shape, **kwargs):
        """hyperprior.py:195-246: the reference's 13-field CompressionOutput (entropy-coded words, shapes, Shannon
        estimates)."""
        from .compression import codec
        nets, fns = self._codec_parts()
        hyp, prior, scale_table = self._tables
        with torch.no_grad():
            return codec.compress_forward(latents.float(), spatial_shape, nets, hyp, prior, scale_table, fns,
                                          vectorize=self.vectorize_encoding, block_encode=self.block_encode,
                                          scale_lower_bound=self.scale_lower_bound, bits_fn=self._shannon_bits)

    def decompress_forward(self, compression_output, device):
        """hyperprior.py:248-274: dequantised latents on `device`."""
        from .compression import codec
        nets, fns = self._codec_parts()
        hyp, prior, scale_table = self._tables
        with torch.no_grad():
            return codec.decompress_forward(compression_output, nets, hyp, prior, scale_table, fns,
                                            n_hyper_channels=self.hyperlatent_filters, device=device,
                                            vectorize=self.vectorize_encoding, block_decode=self.block_encode,
                                            scale_lower_bound=self.scale_lower_bound)

    def forward(self, latents, spatial_shape, defer_rate_join=False, **kwargs):
        """`src/hyperprior.py:277-330`.  The critical path to the Generator is analysis -> quantise -> synthesis_mu ->
        straight-through quantisation; the rate side (both hyperlatent likelihoods, synthesis_std, both latent likelihoods,
        the four entropies) only feeds the loss, so with branch streams on it runs on ops.branch_stream() concurrently
        with whatever the caller launches next (and its backward concurrently with the Generator's).  The caller's stream
        waits for it before this returns unless `defer_rate_join=True` (then the caller must order its first use of the
        *_bpp fields after ops.branch_stream(), as model.py does)."""
        if latents.dtype != torch.float32:
            latents = ops.cast_grad(latents, torch.float32)
        lat_a, lat_b = ops.fork(latents)
        lat_c, lat_d = ops.fork(lat_b)
        lat_e, lat_f = ops.fork(lat_d)

        hyperlatents = self.analysis_net(lat_a)
        hyp_n, hyp_q = ops.fork(hyperlatents)
        noisy_hyperlatents = self._quantize(hyp_n, mode='noise')
        nh_lik, nh_dec = ops.fork(noisy_hyperlatents)
        quantized_hyperlatents = self._quantize(hyp_q, mode='quantize')
        qh_lik, qh_dec = ops.fork(quantized_hyperlatents)

        hyperlatents_decoded = nh_dec if self.training is True else qh_dec
        hd_mu, hd_std = ops.fork(hyperlatents_decoded)

        latent_means = self.synthesis_mu(hd_mu)
        if getattr(self, 'keep_debug', False):          # parity tests: symbols = round(decoded - means)
            self.debug_latent_means = latent_means.detach().float().clone()
            self.debug_latents = latents.detach().float().clone()
        mu_a, mu_b = ops.fork(latent_means)
        mu_c, mu_d = ops.fork(mu_b)
        mu_e, mu_f = ops.fork(mu_d)
        mu_g, mu_h = ops.fork(mu_f)

        latents_decoded = self.quantize_latents_st(lat_f, mu_g)          # end of the critical path

        def rate_side():
            # differential / discrete entropy, hyperlatents
            noisy_hyperlatent_likelihood = self.hyperlatent_likelihood(nh_lik)
            _, noisy_hyperlatent_bpp = self._estimate_entropy(noisy_hyperlatent_likelihood, spatial_shape, want_bits=False)
            quantized_hyperlatent_likelihood = self.hyperlatent_likelihood(qh_lik)
            _, quantized_hyperlatent_bpp = self._estimate_entropy(quantized_hyperlatent_likelihood, spatial_shape, want_bits=False)

            latent_scales = self.synthesis_std(hd_std)
            latent_scales = lower_bound_toward(latent_scales, self.scale_lower_bound)
            sc_a, sc_b = ops.fork(latent_scales)

            # differential entropy, latents (the reference adds noise to the latents irrespective of `means`)
            noisy_latents = self._quantize(lat_c, mode='noise', means=mu_a)
            noisy_latent_likelihood = self.latent_likelihood(noisy_latents, mean=mu_c, scale=sc_a)
            _, noisy_latent_bpp = self._estimate_entropy(noisy_latent_likelihood, spatial_shape, want_bits=False)

            # discrete entropy, latents
            quantized_latents = self._quantize(lat_e, mode='quantize', means=mu_e)
            quantized_latent_likelihood = self.latent_likelihood(quantized_latents, mean=mu_h, scale=sc_b)
            _, quantized_latent_bpp = self._estimate_entropy(quantized_latent_likelihood, spatial_shape, want_bits=False)
            return (noisy_latent_bpp, noisy_hyperlatent_bpp, noisy_latent_bpp + noisy_hyperlatent_bpp,
                    quantized_latent_bpp, quantized_hyperlatent_bpp, quantized_latent_bpp + quantized_hyperlatent_bpp)

        if ops.branch_use(1) and latents.is_cuda:
            main = torch.cuda.current_stream(latents.device)
            s2 = ops.branch_stream(latents.device)
            s2.wait_stream(main)
            for t in (nh_lik, qh_lik, hd_std, lat_c, lat_e, mu_a, mu_c, mu_e, mu_h):
                t.record_stream(s2)
            with torch.cuda.stream(s2):
                bpps = rate_side()
            for t in bpps:
                t.record_stream(main)
            if not defer_rate_join:
                main.wait_stream(s2)
        else:
            bpps = rate_side()

        return HyperInfo(decoded=latents_decoded, latent_nbpp=bpps[0], hyperlatent_nbpp=bpps[1], total_nbpp=bpps[2],
                         latent_qbpp=bpps[3], hyperlatent_qbpp=bpps[4], total_qbpp=bpps[5])

"""HiFIC model stitcher for boxes without the reference checkout (the benchmark, the GPU tests): the call order
and train/eval semantics of the reference's `src/model.py` (`Model.__init__` :37-105, `compression_forward` :119-165,
`discriminator_forward` :167-188, `compression_loss` :201-241, `GAN_loss` :244-260, `compress`/`decompress` :262-344,
`forward` :346-387) over the drop-in modules of this package.  Where the checkout exists, the reference's own `Model`
runs unchanged on the same modules through `inject.patch_reference()`; this file only adds what the kernels need:

  * explicit gradient fan-outs (`ops.fork`) wherever one tensor feeds two consumers, so the sums run in the HIP add
    kernel and bf16/f32 gradients are merged deterministically
  * `device_rate_select=True`: the rate-penalty branch (losses.py:21-25) evaluated on the device (no D2H sync)
  * under torch.distributed the branch uses the global-batch q_bpp (loss/losses.py)
  * loss bookkeeping (`store_loss`, same keys as the reference) gathered in one table instead of per-line `.item()`
"""
from collections import defaultdict, namedtuple

import os

import torch
import torch.nn as nn

from . import ops
from .default_config import ModelModes, ModelTypes
from .network import encoder, generator, discriminator
from . import hyperprior
from .loss import losses
from .loss import perceptual_loss as ps

Intermediates = namedtuple("Intermediates", ["input_image", "reconstruction", "latents_quantized", "n_bpp", "q_bpp"])
Disc_out = namedtuple("disc_out", ["D_real", "D_gen", "D_real_logits", "D_gen_logits"])


# the scalar loss composition as one kernel (ops.LossCombineFn) on the device-rate-select path; HIFIC_FUSED_LOSS=0: torch glue
_FUSED_LOSS = os.environ.get("HIFIC_FUSED_LOSS", "1") not in ("0", "")


class Model(nn.Module):
    def __init__(self, args, logger=None, storage_train=None, storage_test=None, model_mode=ModelModes.TRAINING,
                 model_type=ModelTypes.COMPRESSION, device_rate_select=False, lpips_backbone=None,
                 allow_random_lpips_backbone=False, build_tables=True, lpips_net=None):
        """`lpips_backbone`: path or state_dict of torchvision's pretrained AlexNet (also `args.lpips_backbone`,
        $HIFIC_LPIPS_ALEX_WEIGHTS); without it PerceptualLoss warns loudly (see loss/perceptual_loss.py).
        `lpips_net`: 'alex' (the reference's hard-wired choice, src/model.py:101; default) or 'vgg' (also `args.lpips_net`)."""
        super().__init__()
        self.args, self.logger = args, logger
        self.model_mode, self.model_type = model_mode, model_type
        self.log_interval = args.log_interval
        self.storage_train = storage_train if storage_train is not None else defaultdict(list)
        self.storage_test = storage_test if storage_test is not None else defaultdict(list)
        self.step_counter = 0
        self.device_rate_select = device_rate_select
        self.writeout = True
        if getattr(args, 'use_latent_mixture_model', False):
            raise NotImplementedError("DLMM variant is off by default (default_config.py:89) and out of scope")
        self.image_dims, self.batch_size = args.image_dims, args.batch_size
        # EVALUATION mode (model.py:60-62): the Hyperprior carries the rANS tables (rebuild them with
        # `model.Hyperprior.hyperprior_entropy_model.build_tables()` after loading a checkpoint, as compress.py:61 does)
        self.entropy_code = model_mode == ModelModes.EVALUATION
        C = args.latent_channels
        self.Encoder = encoder.Encoder(self.image_dims, self.batch_size, C=C, channel_norm=args.use_channel_norm)
        self.Generator = generator.Generator(self.image_dims, self.batch_size, C=C,
                                             n_residual_blocks=args.n_residual_blocks,
                                             channel_norm=args.use_channel_norm, sample_noise=args.sample_noise,
                                             noise_dim=args.noise_dim)
        self.Hyperprior = hyperprior.Hyperprior(bottleneck_capacity=C, likelihood_type=args.likelihood_type,
                                                entropy_code=self.entropy_code, lazy_tables=not build_tables)
        self.amortization_models = [self.Encoder, self.Generator, *self.Hyperprior.amortization_models]
        self.use_discriminator = (model_type == ModelTypes.COMPRESSION_GAN and model_mode != ModelModes.EVALUATION)
        self.discriminator_steps, self.Discriminator = 0, None
        if self.use_discriminator:
            assert args.discriminator_steps > 0, 'Must specify nonzero training steps for D!'
            self.discriminator_steps = args.discriminator_steps
            self.Discriminator = discriminator.Discriminator(image_dims=self.image_dims,
                                                             context_dims=args.latent_dims, C=C)
        # LPIPS tensors are unregistered (not in the state_dict, like the reference's DistModel) but follow .to()
        bb = lpips_backbone if lpips_backbone is not None else getattr(args, 'lpips_backbone', None)
        self.perceptual_loss = ps.PerceptualLoss(
            model='net-lin', net=lpips_net or getattr(args, 'lpips_net', None) or 'alex', use_gpu=False,
            allow_random_backbone=allow_random_lpips_backbone,
            backbone_path=bb if isinstance(bb, str) else None,
            backbone_state_dict=bb if isinstance(bb, dict) else None)

    # ---- bookkeeping --------------------------------------------------------------------------------------------
    def store_loss(self, key, loss):
        assert type(loss) == float, 'Call .item() on loss before storage'
        if self.writeout is True:
            (self.storage_train if self.training else self.storage_test)[key].append(loss)

    def _log(self, **scalars):
        """The reference stores its loss terms every `log_interval` steps (model.py:223-239, 253-258, 381)."""
        if self.writeout and (self.step_counter % self.log_interval == 1):
            for key, v in scalars.items():
                self.store_loss(key, float(v))

    # ---- forward pieces -----------------------------------------------------------------------------------------
    def _out_activation(self, reconstruction):
        if self.args.normalize_input_image is True:          # model.py:155-156 (off in the shipped configs)
            return ops.tanh(reconstruction)
        return reconstruction

    def compression_forward(self, x):
        y = self.Encoder(x)
        ops.wait_late_params()     # FusedAdam(overlap_from=...): everything after the Encoder was updated on the optimizer stream
        # the rate side of the hyperprior runs on the branch stream; its join is deferred to the first use of the rates
        hyperinfo = self.Hyperprior(y, spatial_shape=x.size()[2:], defer_rate_join=True)
        lat_gen, lat_disc = ops.fork(hyperinfo.decoded)
        reconstruction = self._out_activation(self.Generator(lat_gen))
        return Intermediates(x, reconstruction, lat_disc, hyperinfo.total_nbpp, hyperinfo.total_qbpp), hyperinfo

    def discriminator_forward(self, intermediates, train_generator):
        """Real/gen batch through D.  Reproduces the reference's pairing quirk: images are cat([real, gen]) while
        the latents are repeat_interleave(latents, 2) (model.py:176-179)."""
        x_gen, x_real = intermediates.reconstruction, intermediates.input_image
        if train_generator is False:
            x_gen = x_gen.detach()
        if x_real.dtype != x_gen.dtype:
            x_real = ops.cast(x_real.contiguous(), x_gen.dtype)
        if hasattr(self.Discriminator, "forward_pairs") and x_gen.is_cuda:
            # cat([x_real, x_gen]) x repeat_interleave(latents, 2) as one gather inside the Discriminator (forward_pairs)
            D_out, D_out_logits = self.Discriminator.forward_pairs(x_real, x_gen, intermediates.latents_quantized.detach())
        else:
            latents = torch.repeat_interleave(intermediates.latents_quantized.detach(), 2, dim=0)
            D_out, D_out_logits = self.Discriminator(torch.cat([x_real, x_gen], dim=0), latents)
        D_real, D_gen = torch.chunk(torch.squeeze(D_out), 2, dim=0)
        D_real_logits, D_gen_logits = torch.chunk(torch.squeeze(D_out_logits), 2, dim=0)
        return Disc_out(D_real, D_gen, D_real_logits, D_gen_logits)

    def distortion_loss(self, x_gen, x_real):
        return ops.MSEFn.apply(x_gen.contiguous(), x_real.contiguous(), 255.0)      # model.py:190-194

    def perceptual_loss_wrapper(self, x_gen, x_real, normalize=True):
        return torch.mean(self.perceptual_loss.forward(x_gen, x_real, normalize=normalize))

    def _compression_terms(self, intermediates):
        """(distortion, LPIPS value per image [B,1,1,1]): the two image-sized reductions of the compression loss."""
        x_real = intermediates.input_image
        x_gen = intermediates.reconstruction
        if self.args.normalize_input_image is True:          # [-1,1] -> [0,1] (model.py:206-209)
            x_real = ops.scale_shift(x_real, 0.5, 0.5)
            x_gen = ops.scale_shift(x_gen, 0.5, 0.5)
        x_gen_mse, x_gen_lpips = ops.fork(x_gen)
        distortion = self.distortion_loss(x_gen_mse, x_real)
        return distortion, self.perceptual_loss.forward(x_gen_lpips, x_real, normalize=True)

    def _fused_loss_ok(self, x):
        """The scalar composition as one device kernel (ops.LossCombineFn): with the device-side rate-penalty rule
        (`device_rate_select`, no `.item()`), on steps that store no loss terms."""
        return bool(self.device_rate_select and x.is_cuda and _FUSED_LOSS
                    and not (self.writeout and (self.step_counter % self.log_interval == 1)))

    def _combined_loss(self, distortion, lp, intermediates, G_loss=None):
        """total = k_M mse + k_P lpips + lambda(q_bpp) n_bpp [+ beta G_loss] (model.py:211-220,373-376; losses.py:8-28)."""
        from .helpers.utils import get_scheduled_params
        from .parallel import allreduce_scalar_mean
        a = self.args
        lam_A = get_scheduled_params(a.lambda_A, a.lambda_schedule, self.step_counter, a.ignore_schedule)
        lam_B = get_scheduled_params(a.lambda_B, a.lambda_schedule, self.step_counter, a.ignore_schedule)
        assert lam_A > lam_B, "Expected lambda_A > lambda_B, got (A) {} <= (B) {}".format(lam_A, lam_B)
        target = get_scheduled_params(a.target_rate, a.target_schedule, self.step_counter, a.ignore_schedule)
        q = allreduce_scalar_mean(intermediates.q_bpp.detach(), None)
        total, _aux = ops.LossCombineFn.apply(distortion, lp.contiguous(), intermediates.n_bpp, q,
                                              G_loss, a.k_M, a.k_P, lam_A, lam_B, target, a.beta)
        return total

    def compression_loss(self, intermediates, hyperinfo):
        distortion, lp = self._compression_terms(intermediates)
        perceptual = torch.mean(lp)
        w_dist, w_perc = self.args.k_M * distortion, self.args.k_P * perceptual
        w_rate, rate_penalty = losses.weighted_rate_loss(
            self.args, total_nbpp=intermediates.n_bpp, total_qbpp=intermediates.q_bpp,
            step_counter=self.step_counter, ignore_schedule=self.args.ignore_schedule,
            device_select=self.device_rate_select)
        w_rd = w_rate + w_dist
        total = w_rd + w_perc
        self._log(rate_penalty=rate_penalty, distortion=distortion, perceptual=perceptual,
                  n_rate=intermediates.n_bpp, q_rate=intermediates.q_bpp,
                  n_rate_latent=hyperinfo.latent_nbpp, q_rate_latent=hyperinfo.latent_qbpp,
                  n_rate_hyperlatent=hyperinfo.hyperlatent_nbpp, q_rate_hyperlatent=hyperinfo.hyperlatent_qbpp,
                  weighted_rate=w_rate, weighted_distortion=w_dist, weighted_perceptual=w_perc, weighted_R_D=w_rd,
                  weighted_compression_loss_sans_G=total)
        return total

    def GAN_loss(self, intermediates, train_generator=False):
        disc_out = self.discriminator_forward(intermediates, train_generator)
        D_loss, G_loss = losses.gan_losses(self.args.gan_loss_type, disc_out)
        if self.writeout and (self.step_counter % self.log_interval == 1):
            self._log(D_gen=torch.mean(disc_out.D_gen), D_real=torch.mean(disc_out.D_real), disc_loss=D_loss,
                      gen_loss=G_loss, weighted_gen_loss=self.args.beta * G_loss)
        return D_loss, G_loss

    def forward(self, x, train_generator=False, return_intermediates=False, writeout=True):
        self.writeout = writeout
        if train_generator is True:
            self.step_counter += 1            # a 'step' is one cycle of G-D training (model.py:351-353)
        branch = ops.branch_streams_on() and x.is_cuda
        self.perceptual_loss.drop_prefetch()
        # (with normalize_input_image the loss sees a rescaled copy of x, which a prefetch on x could never match)
        if branch and ops.branch_use(0) and self.model_mode != ModelModes.EVALUATION \
                and self.args.normalize_input_image is not True:
            # the LPIPS features of the input image depend on nothing the networks compute: start them on the branch
            # stream now, next to the Encoder (perceptual_loss_wrapper below picks them up)
            main = torch.cuda.current_stream(x.device)
            s2 = ops.branch_stream(x.device)
            s2.wait_stream(main)
            x.record_stream(s2)
            with torch.cuda.stream(s2):
                self.perceptual_loss.prefetch_target(x, normalize=True)
        intermediates, hyperinfo = self.compression_forward(x)
        if branch and ops.branch_use(1) and not (self.use_discriminator and ops.branch_use(2)
                                                 and self.model_mode != ModelModes.EVALUATION):
            # rates were produced on the branch stream (Hyperprior.forward, deferred join); below they are used on this one
            torch.cuda.current_stream(x.device).wait_stream(ops.branch_stream(x.device))
        if self.model_mode == ModelModes.EVALUATION:
            # model.py:357-366: no losses, the clamped reconstruction and the quantised rate
            rec = intermediates.reconstruction
            if self.args.normalize_input_image is True:      # model.py:361-363
                rec = ops.scale_shift(rec, 0.5, 0.5)
            return torch.clamp(rec.float(), min=0., max=1.), intermediates.q_bpp
        out = dict()
        inter_c = inter_d = intermediates
        if self.use_discriminator:
            # the reconstruction feeds the compression losses and D: explicit fan-out
            rec_a, rec_b = ops.fork(intermediates.reconstruction)
            inter_c, inter_d = intermediates._replace(reconstruction=rec_a), intermediates._replace(reconstruction=rec_b)
        if self.use_discriminator and ops.branch_use(2) and x.is_cuda:
            # The distortion / LPIPS / rate branch and the Discriminator branch only meet again in the sum below: the
            # first runs on a second stream, concurrently with D.  autograd replays every op's backward on the stream
            # its forward ran on (and synchronises producers with consumers), so the two backward chains overlap too.
            main = torch.cuda.current_stream(x.device)
            s2 = ops.branch_stream(x.device)
            s2.wait_stream(main)
            for t in (inter_c.input_image, inter_c.reconstruction, inter_c.n_bpp, inter_c.q_bpp) + tuple(hyperinfo):
                if torch.is_tensor(t):
                    t.record_stream(s2)
            fused = self._fused_loss_ok(x)
            with torch.cuda.stream(s2):
                if fused:
                    distortion, lp = self._compression_terms(inter_c)
                else:
                    loss = self.compression_loss(inter_c, hyperinfo)
            out['disc'], G_loss = self.GAN_loss(inter_d, train_generator)
            main.wait_stream(s2)
            if fused:
                distortion.record_stream(main); lp.record_stream(main)
                loss = self._combined_loss(distortion, lp, inter_c, G_loss)
            else:
                loss.record_stream(main)
                loss = loss + self.args.beta * G_loss
        elif self._fused_loss_ok(x):
            distortion, lp = self._compression_terms(inter_c)
            G_loss = None
            if self.use_discriminator:
                out['disc'], G_loss = self.GAN_loss(inter_d, train_generator)
            loss = self._combined_loss(distortion, lp, inter_c, G_loss)
        else:
            loss = self.compression_loss(inter_c, hyperinfo)
            if self.use_discriminator:
                out['disc'], G_loss = self.GAN_loss(inter_d, train_generator)
                loss = loss + self.args.beta * G_loss
        out['compression'] = loss
        self._log(weighted_compression_loss=loss)
        return (out, intermediates) if return_intermediates is True else out

    # ---- EVALUATION path (model.py:262-344) -----------------------------------------------------------------------
    def compress(self, x, silent=False):
        """x -> Encoder -> latents -> Hyperprior.compress_forward: the reference's 13-field CompressionOutput."""
        from .helpers import utils
        assert self.model_mode == ModelModes.EVALUATION and (self.training is False), \
            f'Set model mode to {ModelModes.EVALUATION} for compression.'
        spatial_shape = tuple(x.size()[2:])
        ops.wait_late_params()
        with torch.no_grad():
            x = utils.pad_factor(x, x.size()[2:], 2 ** self.Encoder.n_downsampling_layers)
            y = self.Encoder(x.contiguous())
            y = utils.pad_factor(y.float(), y.size()[2:], 2 ** self.Hyperprior.analysis_net.n_downsampling_layers)
            out = self.Hyperprior.compress_forward(y.contiguous(), spatial_shape)
        if silent is False and self.logger is not None:          # model.py:296-309
            attained_hbpp = 32 * len(out.hyperlatents_encoded) / (spatial_shape[0] * spatial_shape[1])
            attained_lbpp = 32 * len(out.latents_encoded) / (spatial_shape[0] * spatial_shape[1])
            self.logger.info('[ESTIMATED]')
            self.logger.info(f'BPP: {out.total_bpp:.3f}')
            self.logger.info(f'HL BPP: {out.hyperlatent_bpp:.3f}')
            self.logger.info(f'L BPP: {out.latent_bpp:.3f}')
            self.logger.info('[ATTAINED]')
            self.logger.info(f'BPP: {attained_hbpp + attained_lbpp:.3f}')
            self.logger.info(f'HL BPP: {attained_hbpp:.3f}')
            self.logger.info(f'L BPP: {attained_lbpp:.3f}')
        return out

    def decompress(self, compression_output):
        """CompressionOutput -> latents (host decode + synthesis nets) -> Generator -> crop to the image size, in [0,1]."""
        assert self.model_mode == ModelModes.EVALUATION and (self.training is False), \
            f'Set model mode to {ModelModes.EVALUATION} for decompression.'
        device = next(self.Generator.parameters()).device
        ops.wait_late_params()
        with torch.no_grad():
            latents_decoded = self.Hyperprior.decompress_forward(compression_output, device=device)
            reconstruction = self._out_activation(self.Generator(latents_decoded.contiguous()))
            H, W = compression_output.spatial_shape
            reconstruction = reconstruction[:, :, :H, :W]
            if self.args.normalize_input_image is True:      # model.py:338-340
                reconstruction = ops.scale_shift(reconstruction.contiguous(), 0.5, 0.5)
            return torch.clamp(reconstruction.float(), min=0., max=1.)

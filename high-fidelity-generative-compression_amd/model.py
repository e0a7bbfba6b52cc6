"""HiFIC model stitcher on the MI355X kernels: mirror of the reference's src/model.py training/validation path
(Model.__init__ :37-105, compression_forward :119-165, discriminator_forward :167-188, distortion_loss :190-194,
compression_loss :201-241, GAN_loss :244-260, forward :346-387), written against the drop-in modules of this
package.  The reference's own Model can also be used unchanged with these modules injected (see inject.py).

Differences that are deliberate and flagged:
  * `device_rate_select=True` evaluates the rate-penalty branch on the device instead of `.item()` (losses.py:21)
  * the bookkeeping `.item()` calls at log steps are kept behind `writeout`
"""
from collections import defaultdict, namedtuple
from functools import partial

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


class Model(nn.Module):
    def __init__(self, args, logger=None, storage_train=None, storage_test=None, model_mode=ModelModes.TRAINING,
                 model_type=ModelTypes.COMPRESSION, device_rate_select=False):
        super().__init__()
        self.args = args
        self.model_mode = model_mode
        self.model_type = model_type
        self.logger = logger
        self.log_interval = args.log_interval
        self.storage_train = storage_train if storage_train is not None else defaultdict(list)
        self.storage_test = storage_test if storage_test is not None else defaultdict(list)
        self.step_counter = 0
        self.device_rate_select = device_rate_select
        self.writeout = True
        if getattr(args, 'use_latent_mixture_model', False):
            raise NotImplementedError("DLMM variant is off by default (default_config.py:89) and out of scope")
        self.image_dims = self.args.image_dims
        self.batch_size = self.args.batch_size
        # EVALUATION mode (model.py:60-62): the Hyperprior carries the rANS tables (rebuild them with
        # `model.Hyperprior.build_tables()` after loading a checkpoint, as compress.py:61 does)
        self.entropy_code = model_mode == ModelModes.EVALUATION

        self.Encoder = encoder.Encoder(self.image_dims, self.batch_size, C=self.args.latent_channels,
                                       channel_norm=self.args.use_channel_norm)
        self.Generator = generator.Generator(self.image_dims, self.batch_size, C=self.args.latent_channels,
                                             n_residual_blocks=self.args.n_residual_blocks,
                                             channel_norm=self.args.use_channel_norm,
                                             sample_noise=self.args.sample_noise, noise_dim=self.args.noise_dim)
        self.Hyperprior = hyperprior.Hyperprior(bottleneck_capacity=self.args.latent_channels,
                                                likelihood_type=self.args.likelihood_type,
                                                entropy_code=self.entropy_code)
        self.amortization_models = [self.Encoder, self.Generator]
        self.amortization_models.extend(self.Hyperprior.amortization_models)

        self.use_discriminator = (self.model_type == ModelTypes.COMPRESSION_GAN
                                  and self.model_mode != ModelModes.EVALUATION)
        if self.use_discriminator:
            assert self.args.discriminator_steps > 0, 'Must specify nonzero training steps for D!'
            self.discriminator_steps = self.args.discriminator_steps
            self.Discriminator = discriminator.Discriminator(image_dims=self.image_dims,
                                                             context_dims=self.args.latent_dims,
                                                             C=self.args.latent_channels)
            self.gan_loss = partial(losses.gan_loss, args.gan_loss_type)
        else:
            self.discriminator_steps = 0
            self.Discriminator = None
        # LPIPS tensors are unregistered (not in the state_dict, like the reference's DistModel) but follow .to()
        self.perceptual_loss = ps.PerceptualLoss(model='net-lin', net='alex', use_gpu=False)

    def store_loss(self, key, loss):
        assert type(loss) == float, 'Call .item() on loss before storage'
        storage = self.storage_train if self.training else self.storage_test
        if self.writeout is True:
            storage[key].append(loss)

    # ------------------------------------------------------------------------------------------------
    def compression_forward(self, x):
        y = self.Encoder(x)
        hyperinfo = self.Hyperprior(y, spatial_shape=x.size()[2:])
        latents_quantized = hyperinfo.decoded
        lat_gen, lat_disc = ops.fork(latents_quantized)
        reconstruction = self.Generator(lat_gen)
        if self.args.normalize_input_image is True:
            raise NotImplementedError("normalize_input_image=True (tanh output) is off in the reference defaults")
        intermediates = Intermediates(x, reconstruction, lat_disc, hyperinfo.total_nbpp, hyperinfo.total_qbpp)
        return intermediates, hyperinfo

    # ---- EVALUATION path (model.py:262-344); GPU wiring not yet run on a device, see DESIGN.md section 7 ----------
    def compress(self, x, silent=True):
        """x -> Encoder -> latents -> Hyperprior.compress_forward: CompressionOutput for `container.save_compressed_format`."""
        from .helpers import utils
        assert self.model_mode == ModelModes.EVALUATION and (self.training is False), \
            f'Set model mode to {ModelModes.EVALUATION} for compression.'
        spatial_shape = tuple(x.size()[2:])
        with torch.no_grad():
            x = utils.pad_factor(x, x.size()[2:], 2 ** self.Encoder.n_downsampling_layers)
            y = self.Encoder(x.contiguous())
            y = utils.pad_factor(y.float(), y.size()[2:], 2 ** self.Hyperprior.analysis_net.n_downsampling_layers)
            return self.Hyperprior.compress_forward(y.contiguous(), spatial_shape)

    def decompress(self, compression_output):
        """CompressionOutput -> latents (host decode + synthesis nets) -> Generator -> crop to the image size, in [0,1]."""
        assert self.model_mode == ModelModes.EVALUATION and (self.training is False), \
            f'Set model mode to {ModelModes.EVALUATION} for decompression.'
        device = next(self.Generator.parameters()).device
        with torch.no_grad():
            latents_decoded = self.Hyperprior.decompress_forward(compression_output, device=device)
            reconstruction = self.Generator(latents_decoded.contiguous())
            if self.args.normalize_input_image is True:
                raise NotImplementedError("normalize_input_image=True (tanh output) is off in the reference defaults")
            H, W = compression_output.spatial_shape
            reconstruction = reconstruction[:, :, :H, :W]
            return torch.clamp(reconstruction.float(), min=0., max=1.)

    def discriminator_forward(self, intermediates, train_generator):
        """Real/gen batch through D.  Reproduces the reference's pairing quirk: images are cat([real, gen]) while
        the latents are repeat_interleave(latents, 2) (model.py:176-179)."""
        x_gen = intermediates.reconstruction
        x_real = intermediates.input_image
        if train_generator is False:
            x_gen = x_gen.detach()
        if x_real.dtype != x_gen.dtype:
            x_real = ops.cast(x_real.contiguous(), x_gen.dtype)
        D_in = torch.cat([x_real, x_gen], dim=0)
        latents = intermediates.latents_quantized.detach()
        latents = torch.repeat_interleave(latents, 2, dim=0)
        D_out, D_out_logits = self.Discriminator(D_in, latents)
        D_out = torch.squeeze(D_out)
        D_out_logits = torch.squeeze(D_out_logits)
        D_real, D_gen = torch.chunk(D_out, 2, dim=0)
        D_real_logits, D_gen_logits = torch.chunk(D_out_logits, 2, dim=0)
        return Disc_out(D_real, D_gen, D_real_logits, D_gen_logits)

    def distortion_loss(self, x_gen, x_real):
        return ops.MSEFn.apply(x_gen.contiguous(), x_real.contiguous(), 255.0)

    def perceptual_loss_wrapper(self, x_gen, x_real, normalize=True):
        lp = self.perceptual_loss.forward(x_gen, x_real, normalize=normalize)
        return torch.mean(lp)

    def compression_loss(self, intermediates, hyperinfo):
        x_real = intermediates.input_image
        x_gen = intermediates.reconstruction
        x_gen_mse, x_gen_lpips = ops.fork(x_gen)
        distortion_loss = self.distortion_loss(x_gen_mse, x_real)
        perceptual_loss = self.perceptual_loss_wrapper(x_gen_lpips, x_real, normalize=True)
        weighted_distortion = self.args.k_M * distortion_loss
        weighted_perceptual = self.args.k_P * perceptual_loss
        weighted_rate, rate_penalty = losses.weighted_rate_loss(
            self.args, total_nbpp=intermediates.n_bpp, total_qbpp=intermediates.q_bpp,
            step_counter=self.step_counter, ignore_schedule=self.args.ignore_schedule,
            device_select=self.device_rate_select)
        weighted_R_D_loss = weighted_rate + weighted_distortion
        weighted_compression_loss = weighted_R_D_loss + weighted_perceptual
        if self.writeout and (self.step_counter % self.log_interval == 1):
            self.store_loss('rate_penalty', float(rate_penalty))
            self.store_loss('distortion', distortion_loss.item())
            self.store_loss('perceptual', perceptual_loss.item())
            self.store_loss('n_rate', intermediates.n_bpp.item())
            self.store_loss('q_rate', intermediates.q_bpp.item())
            self.store_loss('n_rate_latent', hyperinfo.latent_nbpp.item())
            self.store_loss('q_rate_latent', hyperinfo.latent_qbpp.item())
            self.store_loss('n_rate_hyperlatent', hyperinfo.hyperlatent_nbpp.item())
            self.store_loss('q_rate_hyperlatent', hyperinfo.hyperlatent_qbpp.item())
            self.store_loss('weighted_rate', weighted_rate.item())
            self.store_loss('weighted_distortion', weighted_distortion.item())
            self.store_loss('weighted_perceptual', weighted_perceptual.item())
            self.store_loss('weighted_R_D', weighted_R_D_loss.item())
            self.store_loss('weighted_compression_loss_sans_G', weighted_compression_loss.item())
        return weighted_compression_loss

    def GAN_loss(self, intermediates, train_generator=False):
        disc_out = self.discriminator_forward(intermediates, train_generator)
        D_loss, G_loss = losses.gan_losses(self.args.gan_loss_type, disc_out)
        if self.writeout and (self.step_counter % self.log_interval == 1):
            self.store_loss('D_gen', torch.mean(disc_out.D_gen).item())
            self.store_loss('D_real', torch.mean(disc_out.D_real).item())
            self.store_loss('disc_loss', D_loss.item())
            self.store_loss('gen_loss', G_loss.item())
            self.store_loss('weighted_gen_loss', (self.args.beta * G_loss).item())
        return D_loss, G_loss

    def forward(self, x, train_generator=False, return_intermediates=False, writeout=True):
        self.writeout = writeout
        out = dict()
        if train_generator is True:
            self.step_counter += 1
        intermediates, hyperinfo = self.compression_forward(x)
        if self.use_discriminator:
            # the reconstruction feeds the compression losses and D: explicit fan-out
            rec_a, rec_b = ops.fork(intermediates.reconstruction)
            inter_c = intermediates._replace(reconstruction=rec_a)
            inter_d = intermediates._replace(reconstruction=rec_b)
        else:
            inter_c = inter_d = intermediates
        compression_model_loss = self.compression_loss(inter_c, hyperinfo)
        if self.use_discriminator:
            D_loss, G_loss = self.GAN_loss(inter_d, train_generator)
            compression_model_loss = compression_model_loss + self.args.beta * G_loss
            out['disc'] = D_loss
        out['compression'] = compression_model_loss
        if self.writeout and (self.step_counter % self.log_interval == 1):
            self.store_loss('weighted_compression_loss', compression_model_loss.item())
        if return_intermediates is True:
            return out, intermediates
        return out

"""Hot-path constants of the reference's default_config.py (:10-112), restated so the benchmark and tests can build
a HiFIC model where /root/reference does not exist.  Only fields read by Model / losses are kept."""


class ModelTypes(object):
    COMPRESSION = 'compression'
    COMPRESSION_GAN = 'compression_gan'


class ModelModes(object):
    TRAINING = 'training'
    VALIDATION = 'validation'
    EVALUATION = 'evaluation'


class args(object):
    name = 'hific_v0.1'
    silent = True
    n_steps = 1e6
    batch_size = 8
    log_interval = 1000
    save_interval = 50000
    gpu = 0
    # GAN
    discriminator_steps = 0
    model_mode = ModelModes.TRAINING
    sample_noise = False
    noise_dim = 32
    # architecture (Table 3a of arXiv:2006.09965)
    latent_channels = 220
    n_residual_blocks = 9
    lambda_B = 2 ** (-4)
    k_M = 0.075 * 2 ** (-5)
    k_P = 1.
    beta = 0.15
    use_channel_norm = True
    likelihood_type = 'gaussian'
    normalize_input_image = False
    crop_size = 256
    image_dims = (3, 256, 256)
    latent_dims = (latent_channels, 16, 16)
    learning_rate = 1e-4
    weight_decay = 1e-6
    lambda_schedule = dict(vals=[2., 1.], steps=[50000])
    lr_schedule = dict(vals=[1., 0.1], steps=[500000])
    target_schedule = dict(vals=[0.20 / 0.14, 1.], steps=[50000])
    ignore_schedule = False
    regime = 'low'
    target_rate_map = dict(low=0.14, med=0.3, high=0.45)
    lambda_A_map = dict(low=2 ** 1, med=2 ** 0, high=2 ** (-1))
    target_rate = target_rate_map[regime]
    lambda_A = lambda_A_map[regime]
    use_latent_mixture_model = False
    mixture_components = 4
    latent_channels_DLMM = 64


class mse_lpips_args(args):
    model_type = ModelTypes.COMPRESSION


class hific_args(args):
    model_type = ModelTypes.COMPRESSION_GAN
    gan_loss_type = 'non_saturating'
    discriminator_steps = 1
    sample_noise = False


def make_args(base=mse_lpips_args, regime='low', **overrides):
    """dict(class attributes) -> Struct, with target_rate / lambda_A bound to `regime` (train.py:263-271)."""
    from .helpers.utils import Struct
    d = {}
    for klass in reversed(base.__mro__):
        d.update({k: v for k, v in vars(klass).items() if not k.startswith('__')})
    d['regime'] = regime
    d['target_rate'] = args.target_rate_map[regime]
    d['lambda_A'] = args.lambda_A_map[regime]
    d.update(overrides)
    return Struct(**d)

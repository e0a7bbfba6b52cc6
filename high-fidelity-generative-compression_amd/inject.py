"""Drop-in injection: make the *reference's own* `src.model.Model` (and anything else importing `src.network.*`,
`src.normalisation.channel`, `src.hyperprior`, `src.loss.perceptual_similarity.perceptual_loss`) build the MI355X
modules of this package instead of its PyTorch ones.  The seam is module construction inside Model.__init__
(reference src/model.py:68-105); constructors, attribute names and state_dict keys are identical, so reference
checkpoints load unchanged.

    import hific_amd.inject as inject
    inject.patch_reference()            # after the reference's `src` package is importable
    from src.model import Model         # reference code, MI355X kernels underneath
"""
import importlib


def patch_reference():
    from . import hyperprior as hp
    from .network import encoder, generator, discriminator, hyper
    from .normalisation import channel
    from .compression import hyperprior_model
    from .loss import perceptual_loss

    targets = {
        "src.network.encoder": {"Encoder": encoder.Encoder},
        "src.network.generator": {"Generator": generator.Generator, "ResidualBlock": generator.ResidualBlock},
        "src.network.discriminator": {"Discriminator": discriminator.Discriminator},
        "src.network.hyper": {"HyperpriorAnalysis": hyper.HyperpriorAnalysis,
                              "HyperpriorSynthesis": hyper.HyperpriorSynthesis},
        "src.normalisation.channel": {"ChannelNorm2D": channel.ChannelNorm2D,
                                      "ChannelNorm2D_wrap": channel.ChannelNorm2D_wrap},
        "src.hyperprior": {"Hyperprior": hp.Hyperprior, "CodingModel": hp.CodingModel},
        "src.compression.hyperprior_model": {"HyperpriorDensity": hyperprior_model.HyperpriorDensity},
        "src.loss.perceptual_similarity.perceptual_loss": {"PerceptualLoss": perceptual_loss.PerceptualLoss},
    }
    patched = []
    for modname, attrs in targets.items():
        mod = importlib.import_module(modname)
        for name, obj in attrs.items():
            setattr(mod, name, obj)
            patched.append(f"{modname}.{name}")
    return patched

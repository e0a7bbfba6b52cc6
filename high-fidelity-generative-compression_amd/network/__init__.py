from . import encoder, generator, hyper, discriminator  # noqa: F401

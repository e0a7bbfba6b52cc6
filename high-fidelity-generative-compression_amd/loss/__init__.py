from . import losses, perceptual_loss  # noqa: F401

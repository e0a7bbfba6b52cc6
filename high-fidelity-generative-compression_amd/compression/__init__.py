from . import hyperprior_model  # noqa: F401

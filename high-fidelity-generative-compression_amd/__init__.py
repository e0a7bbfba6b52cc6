"""hific_amd — MI355X-native (gfx950) HiFIC forward/backward hot path.

Hand-written HIP kernels behind a C-ABI (libhific_hip.so, declared in include/hific_hip.h) wrapped as drop-in
nn.Module replacements for the reference's src/network, src/normalisation, src/hyperprior and the LPIPS loss.
Import name: `hific_amd` (the on-disk directory keeps the project's hyphenated name).
"""
from . import lib  # noqa: F401  (raises if the HIP library is not built: there is no fallback)
from . import ops  # noqa: F401
from .ops import (set_compute_dtype, get_compute_dtype, set_exact_index, exact_index_on,  # noqa: F401
                  set_exact_reconstruction, exact_reconstruction_on, set_exact_training, exact_training_on)
from . import normalisation, network, compression, hyperprior, loss, helpers, default_config, model, graph  # noqa: F401
from .model import Model  # noqa: F401

__version__ = "0.1.0"

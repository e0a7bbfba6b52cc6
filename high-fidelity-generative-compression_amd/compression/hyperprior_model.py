"""Drop-in for the *density* half of the reference's src/compression/hyperprior_model.py (HyperpriorDensity,
:252-387): per-channel 1-3-3-3-1 monotone MLP CDF, likelihood = |sigmoid(s*u) - sigmoid(s*l)| lower-bounded at
1e-9.  Same constructor and parameter names (H_k, a_k, b_k).  The rANS entropy coder in the same reference file
(HyperpriorEntropyModel) stays on the host and is out of scope (row 12 / §8f)."""
import numpy as np
import torch
import torch.nn as nn

from .. import ops

MIN_LIKELIHOOD = 1e-9
MAX_LIKELIHOOD = 1e3


class HyperpriorDensity(nn.Module):
    def __init__(self, n_channels, init_scale=10., filters=(3, 3, 3), min_likelihood=MIN_LIKELIHOOD,
                 max_likelihood=MAX_LIKELIHOOD, **kwargs):
        super().__init__(**kwargs)
        self.init_scale = float(init_scale)
        self.filters = tuple(int(f) for f in filters)
        if self.filters != (3, 3, 3):
            raise NotImplementedError("the HIP factorised-prior kernel is specialised for filters=(3,3,3)")
        self.min_likelihood = float(min_likelihood)
        self.max_likelihood = float(max_likelihood)
        self.n_channels = n_channels
        self.dtype = torch.float32
        filters = (1,) + self.filters + (1,)
        scale = self.init_scale ** (1 / (len(self.filters) + 1))
        # initialisation (hyperprior_model.py:284-303): H = log(expm1(1/scale/f_{k+1})), a = 0, b ~ U(-.5,.5)
        for k in range(len(self.filters) + 1):
            H_init = np.log(np.expm1(1 / scale / filters[k + 1]))
            H_k = nn.Parameter(torch.ones((n_channels, filters[k + 1], filters[k])))
            torch.nn.init.constant_(H_k, H_init)
            self.register_parameter('H_{}'.format(k), H_k)
            a_k = nn.Parameter(torch.zeros((n_channels, filters[k + 1], 1)))
            self.register_parameter('a_{}'.format(k), a_k)
            b_k = nn.Parameter(torch.zeros((n_channels, filters[k + 1], 1)))
            torch.nn.init.uniform_(b_k, -0.5, 0.5)
            self.register_parameter('b_{}'.format(k), b_k)

    def _params(self):
        return ([getattr(self, f'H_{k}') for k in range(4)] + [getattr(self, f'a_{k}') for k in range(4)] +
                [getattr(self, f'b_{k}') for k in range(4)])

    def likelihood(self, x, collapsed_format=False, **kwargs):
        """x: (N,C,H,W) float32 -> likelihood (N,C,H,W)."""
        if collapsed_format:
            raise NotImplementedError("collapsed_format is only used by the host entropy coder (out of scope)")
        return ops.FactorizedLikFn.apply(x.contiguous(), self.min_likelihood, *self._params())

    def forward(self, x, **kwargs):
        return self.likelihood(x)

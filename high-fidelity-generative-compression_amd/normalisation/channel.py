"""Drop-in for the reference's src/normalisation/channel.py (ChannelNorm2D, ChannelNorm2D_wrap).

Same constructor, same parameters (`gamma`, `beta` shaped (1,C,1,1)); forward runs the fused HIP kernel
(csrc/norm.hip).  `fuse_relu=True` folds the ReLU that follows the norm in the Encoder / Generator into the
same kernel (the reference applies nn.ReLU as the next module: src/network/encoder.py:56-93).
"""
import torch
import torch.nn as nn

from .. import ops


def ChannelNorm2D_wrap(input_channels, momentum=0.1, affine=True, track_running_stats=False, **kwargs):
    return ChannelNorm2D(input_channels, momentum=momentum, affine=affine,
                         track_running_stats=track_running_stats, **kwargs)


class ChannelNorm2D(nn.Module):
    """Per-pixel normalisation over the channel dim with unbiased variance, eps=1e-3 (channel.py:29-59)."""

    def __init__(self, input_channels, momentum=0.1, eps=1e-3, affine=True, fuse_relu=False, **kwargs):
        super().__init__()
        self.momentum = momentum
        self.eps = eps
        self.affine = affine
        self.fuse_relu = fuse_relu
        self.input_channels = input_channels
        if affine is True:
            self.gamma = nn.Parameter(torch.ones(1, input_channels, 1, 1))
            self.beta = nn.Parameter(torch.zeros(1, input_channels, 1, 1))
        else:
            self.register_buffer("gamma", torch.ones(1, input_channels, 1, 1), persistent=False)
            self.register_buffer("beta", torch.zeros(1, input_channels, 1, 1), persistent=False)

    def forward(self, x, resid=None):
        """`resid`: added to the normalised output in the same kernel (ResidualBlock: generator.py:44)."""
        prod = self.__dict__.get("_bias_producer")
        prev_bias = prod.bias if (prod is not None and prod.bias is not None and prod.bias_grad_in_norm) else None
        if resid is not None and (resid.dtype != x.dtype or resid.shape != x.shape):
            return ops.add(ops.channel_norm(x, self.gamma, self.beta, self.eps, relu=self.fuse_relu, prev_bias=prev_bias), resid)
        return ops.channel_norm(x, self.gamma, self.beta, self.eps, relu=self.fuse_relu, prev_bias=prev_bias, resid=resid)


def fuse_bias_grad(conv, norm):
    """`norm` is the ONLY consumer of `conv`'s output (every conv -> ChannelNorm pair of encoder.py:56-93 and
    generator.py:28-42,98-137): the bias gradient of `conv` is sum_{n,h,w} of the norm's input gradient, which the norm's
    backward kernel already has in registers - it writes it, and the conv skips its own reduction pass (two launches over
    an 8 MB tensor per residual-block conv).  No-op for the InstanceNorm fallback.  The reference is held in the norm's
    __dict__ (not as a sub-module: the state_dict layout must stay the reference's)."""
    import os
    if not isinstance(norm, ChannelNorm2D) or conv.bias is None or os.environ.get("HIFIC_FUSE_BIAS_GRAD", "1") == "0":
        return
    norm.__dict__["_bias_producer"] = conv
    conv.bias_grad_in_norm = True

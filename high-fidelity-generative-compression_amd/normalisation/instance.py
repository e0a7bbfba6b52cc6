"""InstanceNorm2D_wrap (reference src/normalisation/instance.py:7-15): `nn.InstanceNorm2d(affine, no running stats)`.

Only reachable with `use_channel_norm=False`; every BASELINE configuration uses ChannelNorm (default_config.py:62),
so this row is NOT a HIP kernel target: it is the documented PyTorch fallback of the package -
the one module whose arithmetic runs in ATen (MIOpen/native instance norm on the device) - so that a model built
with `use_channel_norm=False` constructs, trains and loads reference checkpoints (`weight` / `bias` keys).
"""
import torch.nn as nn


import torch
import torch.nn.functional as F


class _InstanceNorm2dAnyDtype(nn.InstanceNorm2d):
    """nn.InstanceNorm2d with float32 parameters applied to float32 or bfloat16 activations (bf16 compute mode keeps
    activations in bf16 between the HIP kernels).  `fuse_relu` applies the ReLU that the reference places as the next
    module (encoder.py:56-93, generator.py:34): this package's Encoder/Generator fold that ReLU into their norm."""

    def __init__(self, *args, fuse_relu=False, **kwargs):
        super().__init__(*args, **kwargs)
        self.fuse_relu = fuse_relu

    def forward(self, x):
        y = super().forward(x.float()).to(x.dtype) if x.dtype != torch.float32 else super().forward(x)
        return F.relu(y) if self.fuse_relu else y


def InstanceNorm2D_wrap(input_channels, momentum=0.1, affine=True, track_running_stats=False, fuse_relu=False,
                        **kwargs):
    return _InstanceNorm2dAnyDtype(input_channels, momentum=momentum, affine=affine,
                                   track_running_stats=track_running_stats, fuse_relu=fuse_relu)

"""Parameter holders that initialise exactly like torch.nn.Conv2d / ConvTranspose2d (same RNG consumption, same
state_dict keys) but whose forward runs the gfx950 implicit-GEMM kernels (csrc/gconv.hip)."""
import os

import torch
import torch.nn as nn

from .. import ops, lib


class HipConv2d(nn.Conv2d):
    """Conv2d with explicit (top, left, bottom, right) padding of mode 'zeros' | 'reflect' folded into the kernel's
    tile loader (never materialised), optional fused activation, optional float32 output in bf16 mode."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, pads=(0, 0, 0, 0), pad_mode="zeros",
                 act=None, out_f32=False, bias=True):
        super().__init__(in_channels, out_channels, kernel_size, stride=stride, padding=0, bias=bias)
        self.pads = tuple(int(p) for p in pads)
        self.hip_pad_mode = lib.PAD_REFLECT if pad_mode == "reflect" else lib.PAD_ZERO
        self.act = act
        self.out_f32 = out_f32
        self.exact_index_chain = False     # set by the owner: this layer feeds the floor() of the latent indices
        self.exact_recon = False           # set by the owner (Generator): split-bf16 forward under ops.set_exact_reconstruction
        self.bias_grad_in_norm = False     # set by normalisation.channel.fuse_bias_grad: the norm behind this layer owns db

    def forward(self, x):
        return ops.conv2d(x, self.weight, self.bias, stride=self.stride[0], pads=self.pads,
                          pad_mode=self.hip_pad_mode, act=self.act, out_f32=self.out_f32,
                          exact=_exact_mode(self, x), bias_grad=not self.bias_grad_in_norm)

    def extra_repr(self):
        return super().extra_repr() + f", pads(t,l,b,r)={self.pads}, hip_act={self.act}"


class HipConvTranspose2d(nn.ConvTranspose2d):
    """ConvTranspose2d as stride-phase sub-convolutions (no zero insertion), optional fused activation."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, output_padding=0, act=None,
                 out_f32=False):
        super().__init__(in_channels, out_channels, kernel_size, stride=stride, padding=padding,
                         output_padding=output_padding)
        self.act = act
        self.out_f32 = out_f32
        self.exact_index_chain = False
        self.exact_recon = False
        self.bias_grad_in_norm = False

    def forward(self, x):
        return ops.conv_transpose2d(x, self.weight, self.bias, self.stride[0], self.padding[0],
                                    self.output_padding[0], act=self.act, out_f32=self.out_f32,
                                    exact=_exact_mode(self, x), bias_grad=not self.bias_grad_in_norm)


# hyper nets: the 5x5 stride-2 layers (generic kernel) in the pair layout of the native split kernel (ops.SPLIT_PAIR)
_PAIR_HYPER = os.environ.get("HIFIC_EXACT_PAIR_HYPER", "1") not in ("0", "")


_warned_recon_big = False


def _exact_mode(m, x=None):
    """False (plain bf16), True (split operands over 3C channels) or "pair" (native split kernel) for layer m's forward;
    "recon" / "recon_pair" when the layer is exact ONLY as a Generator layer under the exact-training / -reconstruction options
    (ops.Conv2dFn then saves the bf16 image of its float32 input and runs the bf16 backward; a layer of the exact-index chain
    keeps its float32 backward under HIFIC_EXACT_FUSED=0 - ADVICE round 5)."""
    chain = m.exact_index_chain and ops.exact_index_on()
    # exact-reconstruction option: the Generator's layers in no-grad forwards (ops.set_exact_reconstruction)
    # ... and, with ops.set_exact_training, in training forwards too (float32 activations through the autograd graph)
    recon = m.exact_recon and ((ops.exact_reconstruction_on() and not torch.is_grad_enabled()) or ops.exact_training_on())
    if recon and not chain:
        # the split images are addressed with 32-bit element offsets: beyond that (a 60-channel plane of ~11.9 MP) this
        # forward degrades to the plain bf16 path instead of failing
        global _warned_recon_big
        h, w = (int(x.shape[-2]), int(x.shape[-1])) if x is not None else (1, 1)
        if not ops.exact_chain_fits([(m.in_channels, h, w)]):
            if not _warned_recon_big:
                import warnings
                warnings.warn("exact-reconstruction option: input plane beyond the split-image limit, plain bf16 forward")
                _warned_recon_big = True
            return False
    if not (chain or recon):
        return False
    pair = _PAIR_HYPER and ops.exact_pair_on() and m.stride[0] == 2 and m.in_channels >= 32
    if recon and not chain:
        return "recon_pair" if pair else "recon"
    return "pair" if pair else True


def mark_exact_reconstruction(module, on=True):
    """Every conv of `module` (the Generator) runs its NO-GRAD forward with split-bf16 operands when
    ops.set_exact_reconstruction(True) is in effect (bf16 compute mode)."""
    for m in module.modules():
        if isinstance(m, (HipConv2d, HipConvTranspose2d)):
            m.exact_recon = bool(on)
    return module


def mark_exact_index_chain(module, on=True):
    """Every conv of `module` runs its forward with split-bf16 operands in bf16 mode (ops.set_exact_index): the
    Encoder, the hyper-analysis net and the mean synthesis net - the chain behind floor(y - mu + 0.5)
    (src/hyperprior.py:68-74,108-122)."""
    for m in module.modules():
        if isinstance(m, (HipConv2d, HipConvTranspose2d)):
            m.exact_index_chain = bool(on)
    return module

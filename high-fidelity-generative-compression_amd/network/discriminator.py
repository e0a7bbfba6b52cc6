"""Drop-in for the reference's src/network/discriminator.py (Discriminator): same constructor and the same
state_dict keys (`context_conv.*`, `conv{1-4}.{bias,weight_orig,weight_u,weight_v}`, `conv_out.*`).

Spectral norm follows torch.nn.utils.spectral_norm (one power iteration per training-mode forward, in place on the
`weight_u`/`weight_v` buffers, eps 1e-12, sigma = u.(W v), weight = weight_orig/sigma) but runs in csrc/elementwise.hip;
1/sigma is folded into the conv's weight-packing pass.  The nearest x16 upsample + concat (discriminator.py:36,75-77)
is a single gather kernel."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops, lib
from .layers import HipConv2d


class SNConv2d(nn.Module):
    """Conv2d(k=4, s=2, reflect pad 1) under spectral norm, parameter layout of torch's legacy spectral_norm hook."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, pad, act="leaky_relu", eps=1e-12):
        super().__init__()
        conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=0)   # same init as reference
        weight = conv.weight
        self.bias = conv.bias
        self.weight_orig = nn.Parameter(weight.data)
        # torch.nn.utils.spectral_norm: u ~ normalize(N(0,1)^K), v ~ normalize(N(0,1)^(C*R*S)), drawn in this order
        h = weight.shape[0]
        w = weight.numel() // h
        with torch.no_grad():
            u = F.normalize(weight.new_empty(h).normal_(0, 1), dim=0, eps=eps)
            v = F.normalize(weight.new_empty(w).normal_(0, 1), dim=0, eps=eps)
        self.register_buffer("weight_u", u)
        self.register_buffer("weight_v", v)
        self.stride, self.pad, self.act, self.eps = stride, pad, act, eps

    def forward(self, x, sig=None):
        """`sig`: [sigma, 1/sigma] of this forward when the owner already ran the power iteration (Discriminator.forward
        batches the four layers: ops.spectral_norm_power_iteration_batch)."""
        if sig is None:
            with torch.no_grad():
                sig = ops.spectral_norm_power_iteration(self.weight_orig, self.weight_u, self.weight_v,
                                                        do_iter=self.training, eps=self.eps)
        geom = (self.stride, self.pad, self.pad, self.pad, self.pad, lib.PAD_REFLECT)
        return ops.SNConv2dFn.apply(x.contiguous(), self.weight_orig, self.bias, self.weight_u, self.weight_v, sig,
                                    geom, self.act, False)


class Discriminator(nn.Module):
    def __init__(self, image_dims, context_dims, C, spectral_norm=True):
        super().__init__()
        if spectral_norm is not True:
            raise NotImplementedError("only the spectral-norm discriminator (reference default) has kernels")
        self.image_dims = image_dims
        self.context_dims = context_dims
        im_channels = self.image_dims[0]
        kernel_dim = 4
        context_C_out = 12
        filters = (64, 128, 256, 512)
        self.context_conv = HipConv2d(C, context_C_out, 3, stride=1, pads=(1, 1, 1, 1), pad_mode="reflect",
                                      act="leaky_relu")
        self.upsample_factor = 16
        self.conv1 = SNConv2d(im_channels + context_C_out, filters[0], kernel_dim, 2, 1)
        self.conv2 = SNConv2d(filters[0], filters[1], kernel_dim, 2, 1)
        self.conv3 = SNConv2d(filters[1], filters[2], kernel_dim, 2, 1)
        self.conv4 = SNConv2d(filters[2], filters[3], kernel_dim, 2, 1)
        self.conv_out = HipConv2d(filters[3], 1, 1, stride=1, out_f32=True)

    def forward(self, x, y):
        """x: concatenated real/gen images (2B,3,H,W); y: quantised latents (2B,C,H/16,W/16)."""
        y = self.context_conv(y)
        if x.dtype != y.dtype:
            x = ops.cast_grad(x, y.dtype)
        x = ops.UpsampleConcatFn.apply(x.contiguous(), y, self.upsample_factor)
        return self._trunk(x)

    def forward_pairs(self, x_real, x_gen, latents):
        """The same as forward(cat([x_real, x_gen]), repeat_interleave(latents, 2)) - how src/model.py:176-179 calls the
        Discriminator - without materialising either: the context conv runs once per latent and the input gather reads
        latent n >> 1 for image n (ops.UpsamplePairConcatFn).  x_real / x_gen (B,3,H,W), latents (B,C,H/16,W/16)."""
        y = self.context_conv(latents)
        if x_gen.dtype != y.dtype:
            x_gen = ops.cast_grad(x_gen, y.dtype)
        if x_real.dtype != y.dtype:
            x_real = ops.cast(x_real.contiguous(), y.dtype)
        if ops.d1_stage_on() and self._d1_stage_eligible(x_real, y):
            # input gather + first convolution as one autograd node: its backward never forms the 15-channel data gradient
            # (ops.D1StageFn)
            convs = (self.conv1, self.conv2, self.conv3, self.conv4)
            with torch.no_grad():
                sigs = ops.spectral_norm_power_iteration_batch([(c.weight_orig, c.weight_u, c.weight_v) for c in convs],
                                                               do_iter=self.training, eps=convs[0].eps)
            c1 = self.conv1
            geom = (c1.stride, c1.pad, c1.pad, c1.pad, c1.pad, lib.PAD_REFLECT)
            x = ops.D1StageFn.apply(x_real.contiguous(), x_gen.contiguous(), y.contiguous(), self.upsample_factor,
                                    c1.weight_orig, c1.bias, c1.weight_u, c1.weight_v, sigs[0], geom, c1.act)
            return self._trunk(x, sigs=sigs, first=1)
        x = ops.UpsamplePairConcatFn.apply(x_real.contiguous(), x_gen.contiguous(), y, self.upsample_factor)
        return self._trunk(x)

    def _d1_stage_eligible(self, x_real, ctxt):
        """ops.D1StageFn's backward (hific_d1_ctx_grad) exists for the reference's shape only: x16 upsampling, 64 output
        channels, 12 context channels, at least two 16 x 16 cells per side (a 1-row latent is valid in the reference), a plane
        that is a multiple of the cell and a workspace of 4 KiB per cell.  Everything else takes the two-node path
        (UpsamplePairConcatFn + SNConv2d), whose backward handles any shape (the fused node used to be chosen
        unconditionally and then failed in the middle of backward)."""
        B, _, H, W = x_real.shape
        f = self.upsample_factor
        if not (f == 16 and self.conv1.weight_orig.shape[0] == 64 and ctxt.shape[1] == 12 and H % f == 0 and W % f == 0):
            return False
        if H // f < 2 or W // f < 2 or tuple(ctxt.shape[2:]) != (H // f, W // f):
            return False
        return lib.workspace(x_real.device).numel() >= B * (H // f) * (W // f) * 4096

    def _trunk(self, x, sigs=None, first=0):
        # one power iteration per spectral-norm layer and forward (torch.nn.utils.spectral_norm), the four layers per launch
        convs = (self.conv1, self.conv2, self.conv3, self.conv4)
        if sigs is None:
            with torch.no_grad():
                sigs = ops.spectral_norm_power_iteration_batch([(c.weight_orig, c.weight_u, c.weight_v) for c in convs],
                                                               do_iter=self.training, eps=convs[0].eps)
        for c, sg in list(zip(convs, sigs))[first:]:
            x = c(x, sig=sg)
        out_logits = self.conv_out(x).view(-1, 1)
        out = ops.sigmoid(out_logits.detach())
        return out, out_logits

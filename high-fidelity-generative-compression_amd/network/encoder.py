"""Drop-in for the reference's src/network/encoder.py (Encoder): same constructor, same child-module names and
indices (so `Encoder.conv_block{1-5}.1.{weight,bias}`, `.2.{gamma,beta}`, `conv_block_out.1.*` state_dict keys are
identical), forward on the gfx950 kernels.  Reflection pads (encoder.py:46-48) are folded into the conv tile
loader; ChannelNorm + ReLU (encoder.py:58-60) is one fused kernel."""
import warnings

import torch.nn as nn

from ..normalisation import channel, instance
from .layers import HipConv2d, mark_exact_index_chain


_warned_big = False


class Encoder(nn.Module):
    def __init__(self, image_dims, batch_size, activation='relu', C=220, channel_norm=True):
        super().__init__()
        if activation != 'relu':
            raise NotImplementedError("hific_amd Encoder: only activation='relu' (the reference default) has a kernel")
        kernel_dim = 3
        filters = (60, 120, 240, 480, 960)
        im_channels = image_dims[0]
        self.n_downsampling_layers = 4
        norm_kwargs = dict(momentum=0.1, affine=True, track_running_stats=False)
        if channel_norm is True:
            def norm(ch):
                return channel.ChannelNorm2D_wrap(ch, fuse_relu=True, **norm_kwargs)
        else:
            def norm(ch):
                return instance.InstanceNorm2D_wrap(ch, fuse_relu=True, **norm_kwargs)

        # index 0 = the reference's pad module (no parameters), index 3 = its activation module
        self.conv_block1 = nn.Sequential(
            nn.Identity(),
            HipConv2d(im_channels, filters[0], (7, 7), stride=1, pads=(3, 3, 3, 3), pad_mode="reflect"),
            norm(filters[0]),
            nn.Identity(),
        )
        blocks = []
        for i in range(4):
            # asymmetric ReflectionPad2d((left 0, right 1, top 1, bottom 0)) then 3x3 stride 2
            blocks.append(nn.Sequential(
                nn.Identity(),
                HipConv2d(filters[i], filters[i + 1], kernel_dim, stride=2, pads=(1, 0, 0, 1), pad_mode="reflect"),
                norm(filters[i + 1]),
                nn.Identity(),
            ))
        self.conv_block2, self.conv_block3, self.conv_block4, self.conv_block5 = blocks
        # the latents feed the (float32) entropy model: keep them float32 even in bf16 compute mode
        self.conv_block_out = nn.Sequential(
            nn.Identity(),
            HipConv2d(filters[4], C, kernel_dim, stride=1, pads=(1, 1, 1, 1), pad_mode="reflect", out_f32=True),
        )
        for blk in (self.conv_block1, self.conv_block2, self.conv_block3, self.conv_block4, self.conv_block5):
            channel.fuse_bias_grad(blk[1], blk[2])          # conv -> norm: the norm's backward also yields the conv's db
        # the latents are floored into the entropy coder's indices (src/hyperprior.py:68-74): split-bf16 forward
        mark_exact_index_chain(self)

    def _forward_exact_chain(self, x):
        """bf16 mode with the exact-index chain: every conv -> ChannelNorm -> ReLU block is ONE op (ops.ExactConvNormFn) that
        reads the split-bf16 image of its input and emits the nominal bf16 activation plus the next block's split image, so
        the forward values are float32-accurate while the stored activations and the whole backward pass are plain bf16."""
        from .. import ops

        def layout(conv):
            # the stride-2 blocks run on the generic strided kernel, which has the native split form (pair layout: 2C staged
            # channels for the three MFMAs); the 7x7 three-channel layer and the stride-1 960 -> 220 layer (software-pipelined
            # kernel) keep the (hi, lo, hi) x (hi, hi, lo) form over 3C channels
            return ops.SPLIT_PAIR if (ops.exact_pair_on() and conv.stride[0] == 2 and conv.in_channels >= 32) else ops.SPLIT_3C
        blocks = (self.conv_block1, self.conv_block2, self.conv_block3, self.conv_block4, self.conv_block5)
        out = self.conv_block_out[1]
        lays = [layout(blk[1]) for blk in blocks] + [layout(out)]
        x3 = ops.split3_act(x, lays[0])
        h = x
        for i, blk in enumerate(blocks):
            conv, norm = blk[1], blk[2]
            h, x3 = ops.exact_conv_norm(h, x3, conv.weight, conv.bias, conv.stride[0], conv.pads, conv.hip_pad_mode,
                                        norm.gamma, norm.beta, norm.eps, norm.fuse_relu, lays[i], lays[i + 1])
        return ops.conv2d(h, out.weight, out.bias, stride=out.stride[0], pads=out.pads, pad_mode=out.hip_pad_mode,
                          out_f32=True, exact=True, x3=x3)

    def forward(self, x):
        from .. import ops
        import torch
        exact = ops.exact_index_on() and ops.get_compute_dtype() == torch.bfloat16 and x.is_cuda \
            and self.conv_block1[1].exact_index_chain
        if exact and not ops.exact_chain_fits(self._exact_planes(x)):
            # images beyond ~11.9 MP: the split-bf16 operand images outgrow the kernels' 32-bit element offsets
            # (HIFIC_ERR_UNSUPPORTED); the plain bf16 chain still runs - 0.4 % of the latent indices may then differ by
            # one step from a float32 run
            global _warned_big
            if not _warned_big:
                _warned_big = True
                warnings.warn(f"hific_amd Encoder: input {tuple(x.shape)} is too large for the exact-index (split-bf16) "
                              f"chain; this forward runs the plain bf16 chain", RuntimeWarning, stacklevel=2)
            with ops.exact_index_suspended():
                return self._forward_plain(x)
        if exact and isinstance(self.conv_block1[2], channel.ChannelNorm2D) and ops.fused_exact_blocks_on():
            return self._forward_exact_chain(x)
        return self._forward_plain(x)

    def _exact_planes(self, x):
        """(channels, H, W) of the block inputs / outputs the exact chain holds as (hi, lo, hi) images."""
        _, C, H, W = x.shape
        planes = [(C, H, W)]
        for blk in (self.conv_block1, self.conv_block2, self.conv_block3, self.conv_block4, self.conv_block5):
            conv = blk[1]
            pt, pl, pb, pr = conv.pads
            H = (H + pt + pb - conv.kernel_size[0]) // conv.stride[0] + 1
            W = (W + pl + pr - conv.kernel_size[1]) // conv.stride[1] + 1
            planes.append((conv.out_channels, H, W))
        return planes

    def _forward_plain(self, x):
        x = self.conv_block1(x)
        x = self.conv_block2(x)
        x = self.conv_block3(x)
        x = self.conv_block4(x)
        x = self.conv_block5(x)
        return self.conv_block_out(x)

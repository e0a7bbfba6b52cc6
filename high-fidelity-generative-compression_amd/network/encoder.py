"""Drop-in for the reference's src/network/encoder.py (Encoder): same constructor, same child-module names and
indices (so `Encoder.conv_block{1-5}.1.{weight,bias}`, `.2.{gamma,beta}`, `conv_block_out.1.*` state_dict keys are
identical), forward on the gfx950 kernels.  Reflection pads (encoder.py:46-48) are folded into the conv tile
loader; ChannelNorm + ReLU (encoder.py:58-60) is one fused kernel."""
import torch.nn as nn

from ..normalisation import channel, instance
from .layers import HipConv2d, mark_exact_index_chain


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

    def forward(self, x):
        x = self.conv_block1(x)
        x = self.conv_block2(x)
        x = self.conv_block3(x)
        x = self.conv_block4(x)
        x = self.conv_block5(x)
        return self.conv_block_out(x)

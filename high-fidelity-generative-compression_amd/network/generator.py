"""Drop-in for the reference's src/network/generator.py (ResidualBlock, Generator): same constructors, child names
(`conv_block_init.{0,2,3}`, `resblock_{m}.{conv1,conv2,norm1,norm2}`, `upconv_block{1-4}.{0,1}`,
`conv_block_out.1`) and state_dict layout; forward on the gfx950 kernels."""
import torch
import torch.nn as nn

from .. import ops
from ..normalisation import channel, instance
from .layers import HipConv2d, HipConvTranspose2d, mark_exact_reconstruction


def _norm_factory(channel_norm):
    norm_kwargs = dict(momentum=0.1, affine=True, track_running_stats=False)
    if channel_norm is True:
        return lambda ch, relu=False: channel.ChannelNorm2D_wrap(ch, fuse_relu=relu, **norm_kwargs)
    return lambda ch, relu=False: instance.InstanceNorm2D_wrap(ch, fuse_relu=relu, **norm_kwargs)


class ResidualBlock(nn.Module):
    """pad1 -> conv3x3 -> norm -> relu -> pad1 -> conv3x3 -> norm -> (+x)   (generator.py:9-44)."""

    def __init__(self, input_dims, kernel_size=3, stride=1, channel_norm=True, activation='relu'):
        super().__init__()
        if activation != 'relu':
            raise NotImplementedError("hific_amd ResidualBlock: only activation='relu' has a kernel")
        in_channels = input_dims[1]
        norm = _norm_factory(channel_norm)
        p = int((kernel_size - 1) / 2)
        self.conv1 = HipConv2d(in_channels, in_channels, kernel_size, stride=stride, pads=(p, p, p, p), pad_mode="reflect")
        self.conv2 = HipConv2d(in_channels, in_channels, kernel_size, stride=stride, pads=(p, p, p, p), pad_mode="reflect")
        self.norm1 = norm(in_channels, relu=True)
        self.norm2 = norm(in_channels)
        channel.fuse_bias_grad(self.conv1, self.norm1)
        channel.fuse_bias_grad(self.conv2, self.norm2)

    def forward(self, x):
        x_conv, identity_map = ops.fork(x)
        res = self.norm1(self.conv1(x_conv))
        if isinstance(self.norm2, channel.ChannelNorm2D):
            return self.norm2(self.conv2(res), resid=identity_map)         # norm + residual add in one kernel
        res = self.norm2(self.conv2(res))
        return ops.add(res, identity_map)


class Generator(nn.Module):
    def __init__(self, input_dims, batch_size, C=16, activation='relu', n_residual_blocks=8, channel_norm=True,
                 sample_noise=False, noise_dim=32):
        super().__init__()
        if activation != 'relu':
            raise NotImplementedError("hific_amd Generator: only activation='relu' has a kernel")
        kernel_dim = 3
        filters = [960, 480, 240, 120, 60]
        self.n_residual_blocks = n_residual_blocks
        self.sample_noise = sample_noise
        self.noise_dim = noise_dim
        self.n_upsampling_layers = 4
        norm = _norm_factory(channel_norm)
        H0, W0 = input_dims[1:]

        self.conv_block_init = nn.Sequential(
            norm(C),
            nn.Identity(),
            HipConv2d(C, filters[0], (3, 3), stride=1, pads=(1, 1, 1, 1), pad_mode="reflect"),
            norm(filters[0]),
        )
        channel.fuse_bias_grad(self.conv_block_init[2], self.conv_block_init[3])
        if sample_noise is True:
            filters[0] += self.noise_dim            # noise is concatenated to the head (generator.py:105-107, 149-152)
        for m in range(n_residual_blocks):
            self.add_module(f'resblock_{m}', ResidualBlock(input_dims=(batch_size, filters[0], H0, W0),
                                                           channel_norm=channel_norm, activation=activation))
        for i in range(4):
            self.add_module(f'upconv_block{i + 1}', nn.Sequential(
                HipConvTranspose2d(filters[i], filters[i + 1], kernel_dim, stride=2, padding=1, output_padding=1),
                norm(filters[i + 1], relu=True),
                nn.Identity(),
            ))
        for i in range(4):
            blk = getattr(self, f'upconv_block{i + 1}')
            channel.fuse_bias_grad(blk[0], blk[1])
        self.conv_block_out = nn.Sequential(
            nn.Identity(),
            HipConv2d(filters[-1], 3, (7, 7), stride=1, pads=(3, 3, 3, 3), pad_mode="reflect"),
        )
        # EVALUATION option (ops.set_exact_reconstruction): no-grad forwards with split-bf16 operands, float32 activations
        mark_exact_reconstruction(self)

    def _draw_noise(self, shape):
        return torch.randn(shape)

    def forward(self, x):
        head = self.conv_block_init(x)
        if self.sample_noise is True:
            # same draw as the reference: host RNG, then moved to the head's device / dtype (generator.py:149-152)
            B, C, H, W = tuple(head.size())
            z = self._draw_noise((B, self.noise_dim, H, W)).to(head)
            head = torch.cat((head, z), dim=1)
        head_res, head_skip = ops.fork(head)
        x = head_res
        for m in range(self.n_residual_blocks):
            x = getattr(self, f'resblock_{m}')(x)
        x = ops.add(x, head_skip)
        x = self.upconv_block1(x)
        x = self.upconv_block2(x)
        x = self.upconv_block3(x)
        x = self.upconv_block4(x)
        return self.conv_block_out(x)

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

    # ---- exact chain (round 6): float32-accurate forward values on a plain-bf16 autograd graph ---------------------------
    def _exact_chain_ok(self, x):
        """bf16 mode, ChannelNorm variant, no noise concat, and the exact-training option on - or the exact-reconstruction
        option in a no-grad forward (Model.decompress / EVALUATION)."""
        if not (x.is_cuda and ops.get_compute_dtype() == torch.bfloat16 and self.sample_noise is not True):
            return False
        if not (isinstance(self.conv_block_init[3], channel.ChannelNorm2D) and self.conv_block_out[1].exact_recon):
            return False
        if not ops.exact_generator_fused_on():
            return False
        on = ops.exact_training_on() or (ops.exact_reconstruction_on() and not torch.is_grad_enabled())
        if not on:
            return False
        N, C, H, W = x.shape
        planes = [(960, H, W), (480, 2 * H, 2 * W), (240, 4 * H, 4 * W), (120, 8 * H, 8 * W), (60, 16 * H, 16 * W)]
        return ops.exact_chain_fits(planes)

    def _forward_exact_chain(self, x):
        """Every conv -> ChannelNorm block as ONE op (ops.ExactConvNormFn, as in the Encoder's exact-index chain): the
        contraction reads the split-bf16 image of its input (x*w ~ xh*wh + xl*wh + xh*wl, float32 accumulate), the norm kernel
        emits the nominal bf16 activation (what autograd stores and the backward pass reads), the next block's split image and
        bf16(z) for its own backward; the residual stream and the head skip are added from the split images (hi + lo).  The
        autograd graph - and so the whole backward pass - is the plain bf16 one; only the forward VALUES are float32-accurate
        (reconstruction within north_star's 1e-3 of the reference Generator, generator.py:145-168)."""
        S3, SP = ops.SPLIT_3C, ops.SPLIT_PAIR

        def up_layout(conv):
            return SP if (ops.exact_pair_on() and conv.in_channels >= 32) else S3
        ups = [getattr(self, f'upconv_block{i + 1}') for i in range(4)]
        up_lay = [up_layout(u[0]) for u in ups]
        n0, conv0, norm0 = self.conv_block_init[0], self.conv_block_init[2], self.conv_block_init[3]
        h = n0(x)                                               # float32 norm of the (float32) decoded latents
        if h.dtype != torch.float32:
            h = ops.cast_grad(h, torch.float32)
        h3 = ops.split3_act(h, S3)
        head, head3 = ops.exact_conv_norm(h, h3, conv0.weight, conv0.bias, 1, conv0.pads, conv0.hip_pad_mode, norm0.gamma,
                                          norm0.beta, norm0.eps, norm0.fuse_relu, S3, S3)
        head_res, head_skip = ops.fork(head)
        h, h3 = head_res, head3
        for m in range(self.n_residual_blocks):
            blk = getattr(self, f'resblock_{m}')
            c1, c2, n1, n2 = blk.conv1, blk.conv2, blk.norm1, blk.norm2
            h_conv, h_id = ops.fork(h)
            r, r3 = ops.exact_conv_norm(h_conv, h3, c1.weight, c1.bias, 1, c1.pads, c1.hip_pad_mode, n1.gamma, n1.beta, n1.eps,
                                        n1.fuse_relu, S3, S3)
            h, h3 = ops.exact_conv_norm(r, r3, c2.weight, c2.bias, 1, c2.pads, c2.hip_pad_mode, n2.gamma, n2.beta, n2.eps,
                                        n2.fuse_relu, S3, S3, resid=h_id, resid3=h3, lay_res=S3)
        h, h3 = ops.add_split(h, h3, S3, head_skip, head3, S3, up_lay[0])
        for i, u in enumerate(ups):
            ct, nm = u[0], u[1]
            lay_next = up_lay[i + 1] if i + 1 < 4 else S3
            h, h3 = ops.exact_conv_transpose_norm(h, h3, ct.weight, ct.bias, ct.stride[0], ct.padding[0], ct.output_padding[0],
                                                  nm.gamma, nm.beta, nm.eps, nm.fuse_relu, up_lay[i], lay_next)
        out = self.conv_block_out[1]
        return ops.conv2d(h, out.weight, out.bias, stride=1, pads=out.pads, pad_mode=out.hip_pad_mode, out_f32=True,
                          exact=True, x3=h3)

    def _plain_fused_ok(self, x):
        """Every conv -> norm pair as one autograd node (ops.ConvNormFn: same kernels, bit-identical, fewer Python-level ops):
        ChannelNorm variant whose norms own the conv bias gradients, no noise concat."""
        if not (x.is_cuda and ops.conv_norm_fused_on() and self.sample_noise is not True):
            return False
        c0, n0 = self.conv_block_init[2], self.conv_block_init[3]
        return isinstance(n0, channel.ChannelNorm2D) and (c0.bias is None or c0.bias_grad_in_norm) \
            and not any(m.exact_index_chain for m in (c0, self.conv_block_out[1]))

    def _forward_plain_fused(self, x):
        cn = ops.conv_norm
        head = cn(self.conv_block_init[0](x), self.conv_block_init[2], self.conv_block_init[3])
        head_res, head_skip = ops.fork(head)
        h = head_res
        for m in range(self.n_residual_blocks):
            blk = getattr(self, f'resblock_{m}')
            h_conv, h_id = ops.fork(h)
            h = cn(cn(h_conv, blk.conv1, blk.norm1), blk.conv2, blk.norm2, resid=h_id)
        h = ops.add(h, head_skip)
        for i in range(4):
            u = getattr(self, f'upconv_block{i + 1}')
            h = cn(h, u[0], u[1])
        return self.conv_block_out(h)

    def forward(self, x):
        if self._exact_chain_ok(x):
            return self._forward_exact_chain(x)
        exact_unfused = self.conv_block_out[1].exact_recon and (
            ops.exact_training_on() or (ops.exact_reconstruction_on() and not torch.is_grad_enabled()))
        if not exact_unfused and self._plain_fused_ok(x):
            return self._forward_plain_fused(x)
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

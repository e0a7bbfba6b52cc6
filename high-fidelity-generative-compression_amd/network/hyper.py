"""Drop-in for the reference's src/network/hyper.py (HyperpriorAnalysis, HyperpriorSynthesis): same constructors and
`conv{1-3}` parameter names.  ReLUs (hyper.py:59-60,91-92) are fused into the conv epilogues; the entropy model
around these nets is float32, so the first conv reads float32 and the last conv writes float32 in bf16 mode."""
import torch.nn as nn

from .layers import HipConv2d, HipConvTranspose2d, mark_exact_index_chain  # noqa: F401


class HyperpriorAnalysis(nn.Module):
    def __init__(self, C=220, N=320, activation='relu'):
        super().__init__()
        if activation != 'relu':
            raise NotImplementedError("only activation='relu' has a kernel")
        self.n_downsampling_layers = 2
        self.conv1 = HipConv2d(C, N, 3, stride=1, pads=(1, 1, 1, 1), pad_mode="zeros", act="relu")
        self.conv2 = HipConv2d(N, N, 5, stride=2, pads=(2, 2, 2, 2), pad_mode="reflect", act="relu")
        self.conv3 = HipConv2d(N, N, 5, stride=2, pads=(2, 2, 2, 2), pad_mode="reflect", out_f32=True)

    def forward(self, x):
        return self.conv3(self.conv2(self.conv1(x)))


class HyperpriorSynthesis(nn.Module):
    def __init__(self, C=220, N=320, activation='relu', final_activation=None):
        super().__init__()
        if activation != 'relu' or final_activation is not None:
            raise NotImplementedError("only activation='relu', final_activation=None (reference usage) have kernels")
        self.final_activation = None
        self.conv1 = HipConvTranspose2d(N, N, 5, stride=2, padding=2, output_padding=1, act="relu")
        self.conv2 = HipConvTranspose2d(N, N, 5, stride=2, padding=2, output_padding=1, act="relu")
        self.conv3 = HipConvTranspose2d(N, C, 3, stride=1, padding=1, out_f32=True)

    def forward(self, x):
        return self.conv3(self.conv2(self.conv1(x)))

"""InstanceNorm2D_wrap (reference src/normalisation/instance.py:7-15).  Only reachable with use_channel_norm=False,
which none of the BASELINE configs use (default_config.py:62); it is not a HIP kernel target."""


def InstanceNorm2D_wrap(input_channels, momentum=0.1, affine=True, track_running_stats=False, **kwargs):
    raise NotImplementedError(
        "hific_amd implements the ChannelNorm path (use_channel_norm=True, the reference default); "
        "InstanceNorm has no HIP kernel and there is no PyTorch fallback on the hot path")

"""The two host helpers of the reference's src/helpers/utils.py that the hot path touches."""


def get_scheduled_params(param, param_schedule, step_counter, ignore_schedule=False):
    """utils.py:64-72: value scaled by vals[idx], idx = first schedule boundary that exceeds step_counter."""
    if ignore_schedule is False:
        vals, steps = param_schedule['vals'], param_schedule['steps']
        assert len(vals) == len(steps) + 1, f'Mispecified schedule! - {param_schedule}'
        idx = len(steps)
        for i, s in enumerate(steps):
            if step_counter < s:
                idx = i
                break
        param = param * vals[idx]
    return param


class Struct:
    def __init__(self, **entries):
        self.__dict__.update(entries)


def pad_factor(input_image, spatial_dims, factor):
    """Reflect-pad (N,C,H,W) at the bottom/right so that H and W are divisible by `factor`
    (src/helpers/utils.py:50-62; used by Model.compress on the image and on the latents)."""
    factor_H, factor_W = (factor, factor) if isinstance(factor, int) else factor
    H, W = spatial_dims[0], spatial_dims[1]
    pad_H = (factor_H - (H % factor_H)) % factor_H
    pad_W = (factor_W - (W % factor_W)) % factor_W
    if pad_H == 0 and pad_W == 0:
        return input_image
    if input_image.is_cuda:                      # hific_pad2d: no ATen arithmetic on the EVALUATION path either
        import torch
        from .. import lib
        x = input_image.contiguous()
        N, C = x.shape[0], x.shape[1]
        y = torch.empty((N, C, H + pad_H, W + pad_W), dtype=x.dtype, device=x.device)
        lib.call("hific_pad2d", x.data_ptr(), y.data_ptr(), N * C, x.shape[2], x.shape[3], 0, 0, pad_H, pad_W, 1,
                 lib.dtype_code(x), lib.stream())
        return y
    import torch.nn.functional as F             # host tensors (tests of the host logic): the reference's own call
    return F.pad(input_image, pad=(0, pad_W, 0, pad_H), mode='reflect')

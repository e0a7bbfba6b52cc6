"""Drop-in for the reference's src/loss/losses.py: scheduled rate penalty and the GAN losses.

weighted_rate_loss reproduces the reference's host-side `.item()` branch (losses.py:21-25) by default; with
`device_select=True` the same rule is evaluated on the device (no D2H sync; used by the benchmark/graph path)."""
import torch

from .. import ops
from ..helpers.utils import get_scheduled_params


def weighted_rate_loss(config, total_nbpp, total_qbpp, step_counter, ignore_schedule=False, device_select=False):
    lambda_A = get_scheduled_params(config.lambda_A, config.lambda_schedule, step_counter, ignore_schedule)
    lambda_B = get_scheduled_params(config.lambda_B, config.lambda_schedule, step_counter, ignore_schedule)
    assert lambda_A > lambda_B, "Expected lambda_A > lambda_B, got (A) {} <= (B) {}".format(lambda_A, lambda_B)
    target_bpp = get_scheduled_params(config.target_rate, config.target_schedule, step_counter, ignore_schedule)
    if device_select:
        rate_penalty = torch.where(total_qbpp.detach() > target_bpp,
                                   torch.full_like(total_qbpp, lambda_A), torch.full_like(total_qbpp, lambda_B))
        return rate_penalty * total_nbpp, rate_penalty
    q = total_qbpp.item()
    rate_penalty = lambda_A if q > target_bpp else lambda_B
    return rate_penalty * total_nbpp, float(rate_penalty)


def _non_saturating_loss(D_real_logits, D_gen_logits, D_real=None, D_gen=None):
    """BCE-with-logits against ones/zeros (losses.py:30-41).  The generated logits feed two losses: explicit fork."""
    gen_a, gen_b = ops.fork(D_gen_logits)
    D_loss_real = ops.BCELogitsFn.apply(D_real_logits.contiguous(), 1.0)
    D_loss_gen = ops.BCELogitsFn.apply(gen_a.contiguous(), 0.0)
    D_loss = D_loss_real + D_loss_gen
    G_loss = ops.BCELogitsFn.apply(gen_b.contiguous(), 1.0)
    return D_loss, G_loss


def gan_loss(gan_loss_type, disc_out, mode='generator_loss'):
    if gan_loss_type != 'non_saturating':
        raise NotImplementedError("only the reference default gan_loss_type='non_saturating' has kernels")
    D_loss, G_loss = _non_saturating_loss(D_real=disc_out.D_real, D_gen=disc_out.D_gen,
                                          D_real_logits=disc_out.D_real_logits, D_gen_logits=disc_out.D_gen_logits)
    return G_loss if mode == 'generator_loss' else D_loss


def gan_losses(gan_loss_type, disc_out):
    """Both losses from one evaluation (the reference calls gan_loss twice on the same Disc_out, model.py:249-250)."""
    if gan_loss_type != 'non_saturating':
        raise NotImplementedError("only gan_loss_type='non_saturating' has kernels")
    return _non_saturating_loss(D_real=disc_out.D_real, D_gen=disc_out.D_gen,
                                D_real_logits=disc_out.D_real_logits, D_gen_logits=disc_out.D_gen_logits)

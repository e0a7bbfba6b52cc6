"""Drop-in for the reference's src/loss/losses.py: scheduled rate penalty and the GAN losses.

weighted_rate_loss reproduces the reference's host-side `.item()` branch (losses.py:21-25) by default; with
`device_select=True` the same rule is evaluated on the device (no D2H sync; used by the benchmark/graph path)."""
import torch

from .. import ops
from ..helpers.utils import get_scheduled_params


def weighted_rate_loss(config, total_nbpp, total_qbpp, step_counter, ignore_schedule=False, device_select=False,
                       process_group=None):
    """losses.py:8-28.  Under data parallelism (torch.distributed initialised, world > 1) the branch is taken on the
    GLOBAL-batch mean of q_bpp (one scalar all-reduce), so every rank applies the same lambda - a single process
    with the global batch would."""
    lambda_A = get_scheduled_params(config.lambda_A, config.lambda_schedule, step_counter, ignore_schedule)
    lambda_B = get_scheduled_params(config.lambda_B, config.lambda_schedule, step_counter, ignore_schedule)
    assert lambda_A > lambda_B, "Expected lambda_A > lambda_B, got (A) {} <= (B) {}".format(lambda_A, lambda_B)
    target_bpp = get_scheduled_params(config.target_rate, config.target_schedule, step_counter, ignore_schedule)
    from ..parallel import allreduce_scalar_mean
    q = allreduce_scalar_mean(total_qbpp.detach(), process_group)
    if device_select:
        rate_penalty = torch.where(q > target_bpp, torch.full_like(q, lambda_A), torch.full_like(q, lambda_B))
        return rate_penalty * total_nbpp, rate_penalty
    rate_penalty = lambda_A if q.item() > target_bpp else lambda_B
    return rate_penalty * total_nbpp, float(rate_penalty)


def _non_saturating_loss(D_real_logits, D_gen_logits, D_real=None, D_gen=None):
    """BCE-with-logits against ones/zeros (losses.py:30-41).  The generated logits feed two losses: explicit fork."""
    gen_a, gen_b = ops.fork(D_gen_logits)
    D_loss_real = ops.BCELogitsFn.apply(D_real_logits.contiguous(), 1.0)
    D_loss_gen = ops.BCELogitsFn.apply(gen_a.contiguous(), 0.0)
    D_loss = D_loss_real + D_loss_gen
    G_loss = ops.BCELogitsFn.apply(gen_b.contiguous(), 1.0)
    return D_loss, G_loss


def _least_squares_loss(D_real=None, D_gen=None, D_real_logits=None, D_gen_logits=None):
    """losses.py:43-50 on D = sigmoid(logits) (discriminator.py:84): the sigmoid and its derivative live inside
    the kernel, so the loss is taken from the logits."""
    gen_a, gen_b = ops.fork(D_gen_logits)
    D_loss_real = ops.LsqSigmoidFn.apply(D_real_logits.contiguous(), 1.0)
    D_loss_gen = ops.LsqSigmoidFn.apply(gen_a.contiguous(), 0.0)
    D_loss = 0.5 * (D_loss_real + D_loss_gen)
    G_loss = 0.5 * ops.LsqSigmoidFn.apply(gen_b.contiguous(), 1.0)
    return D_loss, G_loss


_GAN_LOSSES = {'non_saturating': _non_saturating_loss, 'least_squares': _least_squares_loss}


def gan_losses(gan_loss_type, disc_out):
    """Both losses from one evaluation (the reference calls gan_loss twice on the same Disc_out, model.py:249-250)."""
    if gan_loss_type not in _GAN_LOSSES:
        raise ValueError('Invalid GAN loss')
    return _GAN_LOSSES[gan_loss_type](D_real=disc_out.D_real, D_gen=disc_out.D_gen,
                                      D_real_logits=disc_out.D_real_logits, D_gen_logits=disc_out.D_gen_logits)


def gan_loss(gan_loss_type, disc_out, mode='generator_loss'):
    D_loss, G_loss = gan_losses(gan_loss_type, disc_out)
    return G_loss if mode == 'generator_loss' else D_loss

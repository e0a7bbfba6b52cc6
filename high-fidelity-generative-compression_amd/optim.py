"""Flat parameter arenas + fused Adam for the HiFIC optimizers (reference train.py:287-301: three torch.optim.Adam
with lr 1e-4, betas (0.9, 0.999), eps 1e-8, no weight decay — `amort`, `hyper`, `disc`).

A ParamArena re-points every parameter of a group into one contiguous float32 buffer and gives each a *gradient
slot* in a second buffer.  The backward kernels (weight-gradient finalize, bias/affine reductions) write straight
into the slots, so after `loss.backward()` the whole gradient of a group is one flat tensor:
  * fused Adam is ONE kernel over the arena (csrc/elementwise.hip adam_kernel), no per-tensor launches
  * data-parallel training all-reduces contiguous slices of that tensor (parallel.py), no flatten/copy
Gradient accumulation semantics of autograd are preserved per slot: the first write after `zero_grad()`
overwrites, later writes add (needed for the reference's Discriminator-gradient carry-over).
"""
import torch

from . import ops

_ALIGN = 64   # elements (256 B)


class GradSlot:
    __slots__ = ("grad", "fresh", "arena", "index")

    def __init__(self, grad, arena, index):
        self.grad = grad
        self.fresh = True
        self.arena = arena
        self.index = index

    def take(self):
        """Returns the accumulate flag for the next write (0 = overwrite) and marks the slot written."""
        acc = 0 if self.fresh else 1
        self.fresh = False
        return acc

    def written(self):
        """Called after the kernel that fills this slot has been enqueued (data-parallel bucket bookkeeping)."""
        if self.arena.on_write is not None:
            self.arena.on_write(self)


class ParamArena:
    def __init__(self, params, device=None):
        params = [p for p in params if p.requires_grad]
        assert params, "empty parameter group"
        device = device or params[0].device
        offs, total = [], 0
        for p in params:
            offs.append(total)
            total += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.params = params
        self.offsets = offs
        self.numel = total
        self.flat = torch.zeros(total, dtype=torch.float32, device=device)
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=device)
        self.slots = []
        self.on_write = None
        with torch.no_grad():
            for i, (p, o) in enumerate(zip(params, offs)):
                n = p.numel()
                self.flat[o:o + n].copy_(p.data.reshape(-1))
                p.data = self.flat[o:o + n].view(p.shape)
                gview = self.flat_grad[o:o + n].view(p.shape)
                slot = GradSlot(gview, self, i)
                p._hific_slot = slot
                p.grad = gview
                self.slots.append(slot)

    def zero_grad(self):
        """Marks every slot fresh (next backward overwrites): no memset needed."""
        for s in self.slots:
            s.fresh = True

    def slice_of(self, i):
        o = self.offsets[i]
        return o, self.params[i].numel()


class FusedAdam:
    """torch.optim.Adam semantics (amsgrad=False, weight_decay=0) over one ParamArena."""

    def __init__(self, params, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, device=None):
        self.arena = params if isinstance(params, ParamArena) else ParamArena(list(params), device)
        self.lr, self.betas, self.eps = lr, betas, eps
        self.exp_avg = torch.zeros_like(self.arena.flat)
        self.exp_avg_sq = torch.zeros_like(self.arena.flat)
        self.step_count = 0
        self.grad_scale = 1.0
        self.param_groups = [dict(lr=lr, params=self.arena.params)]

    def step(self):
        self.step_count += 1
        ops.adam_step(self.arena.flat, self.arena.flat_grad, self.exp_avg, self.exp_avg_sq,
                      self.param_groups[0]["lr"], self.betas[0], self.betas[1], self.eps, self.step_count,
                      self.grad_scale)

    def zero_grad(self, set_to_none=False):
        self.arena.zero_grad()

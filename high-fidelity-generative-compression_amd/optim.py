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
    __slots__ = ("grad", "fresh", "arena", "index", "autograd")

    def __init__(self, grad, arena, index):
        self.grad = grad
        self.fresh = True
        self.arena = arena
        self.index = index
        self.autograd = False     # a gradient from autograd's AccumulateGrad is in flight for this slot (see ParamArena)

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
        self.epoch = 0          # bumped by every optimizer step over this arena (ops.WeightPackCache tokens)
        self.late_start, self.late_event = 0, None      # element offset / completion event of a pending optimizer tail
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
                pre, post = self._autograd_hooks(i)
                p.register_hook(pre)
                p.register_post_accumulate_grad_hook(post)

    def _autograd_hooks(self, i):
        """Parameters whose gradient is produced by ATen ops (the InstanceNorm fallback's affine pair, or any user module
        mixed into a group) are accumulated IN PLACE into the slot view by autograd's AccumulateGrad, which knows nothing of
        `fresh`.  The tensor hook sees the incoming gradient BEFORE that accumulation (it is None for every parameter whose
        gradient a kernel wrote into the slot: those return None to autograd): on the first real gradient after zero_grad()
        the slot is cleared and marked written, so the sum lands on zeros, zero_unwritten() keeps it and data-parallel
        buckets count it once the accumulation has happened (post-accumulate hook)."""
        def pre(grad):
            if grad is None:
                return None
            s = self.slots[i]
            if s.fresh:
                s.grad.zero_()
                s.fresh = False
            s.autograd = True
            return None

        def post(p):
            s = self.slots[i]
            if not s.autograd:
                return
            s.autograd = False
            if p.grad is not None and p.grad.data_ptr() != s.grad.data_ptr():
                s.grad.copy_(p.grad)            # autograd installed its own tensor (p.grad had been dropped): bring it home
                p.grad = s.grad
            s.written()
        return pre, post

    def zero_grad(self):
        """Marks every slot fresh (next backward overwrites): no memset needed."""
        for s in self.slots:
            s.fresh = True

    def slice_of(self, i):
        o = self.offsets[i]
        return o, self.params[i].numel()

    def rebind(self):
        """Re-establishes the aliasing `p.data` / `p.grad` -> arena slices for parameters that were re-pointed
        behind the arena's back (any nn.Module._apply: model.cpu() / .to(device) / .float(), as the reference's
        save_model does every epoch, utils.py:116-145).  The parameter's current values win (they are what the user
        sees, e.g. after load_state_dict on a moved model).  Returns the number of parameters that had strayed;
        raises if one now lives on another device."""
        base, gbase = self.flat.data_ptr(), self.flat_grad.data_ptr()
        strayed = 0
        with torch.no_grad():
            for i, (p, o) in enumerate(zip(self.params, self.offsets)):
                n = p.numel()
                want = base + 4 * o
                grad_ok = p.grad is not None and p.grad.data_ptr() == gbase + 4 * o
                if p.data_ptr() == want and grad_ok:
                    continue
                if p.device != self.flat.device:
                    raise RuntimeError(
                        f"ParamArena: parameter {i} {tuple(p.shape)} now lives on {p.device}, the arena on "
                        f"{self.flat.device}; move the model back before optimizer.step() (its weights would "
                        f"otherwise stay frozen copies)")
                if p.data_ptr() != want:
                    self.flat[o:o + n].copy_(p.data.reshape(-1).to(torch.float32))
                    p.data = self.flat[o:o + n].view(p.shape)
                    strayed += 1
                if not grad_ok:
                    # _apply re-points `p.grad.data` IN PLACE, i.e. the slot's own view object, so a backward that
                    # ran after the move wrote its gradient into that stray storage: bring it home, keep `fresh`
                    gview = self.flat_grad[o:o + n].view(p.shape)
                    if p.grad is not None and p.grad.shape == p.shape and not self.slots[i].fresh:
                        gview.copy_(p.grad.to(torch.float32))
                    self.slots[i].grad = gview
                    p.grad = gview
        return strayed

    def zero_unwritten(self):
        """Zeroes the slots that received no gradient since zero_grad() (they still hold an older step's values):
        the optimizer then sees a zero gradient for them, like the reference's opt.zero_grad() (torch 1.6: grads are
        zeroed, not dropped) followed by a backward that does not reach the parameter."""
        lo = hi = None
        for s, o in zip(self.slots, self.offsets):
            if not s.fresh:
                continue
            n = (s.grad.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
            if hi == o:
                hi = o + n
            else:
                if lo is not None:
                    self.flat_grad[lo:hi].zero_()
                lo, hi = o, o + n
        if lo is not None:
            self.flat_grad[lo:hi].zero_()


class FusedAdam:
    """torch.optim.Adam semantics (amsgrad=False, weight_decay=0) over one ParamArena."""

    def __init__(self, params, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, device=None, overlap_from=None):
        self.arena = params if isinstance(params, ParamArena) else ParamArena(list(params), device)
        self.lr, self.betas, self.eps = lr, betas, eps
        self.exp_avg = torch.zeros_like(self.arena.flat)
        self.exp_avg_sq = torch.zeros_like(self.arena.flat)
        # the step count lives on the device (a captured hipGraph of the training step must be replayable: kernel arguments
        # are frozen at capture); `step_count` reads it back
        dev = self.arena.flat.device
        self._step_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        self._bc_dev = torch.zeros(2, dtype=torch.float32, device=dev)
        self.grad_scale = 1.0
        # number of leading parameters updated on the current stream; the rest go to ops.opt_stream() (None: all here).
        # The caller must then use ops.wait_late_params() / synchronize() before touching the tail: hific_amd.Model does
        self.overlap_from = overlap_from
        self.param_groups = [dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False,
                                  params=self.arena.params)]

    def step(self):
        a = self.arena
        a.rebind()
        a.zero_unwritten()
        g = self.param_groups[0]
        hyper = (g["lr"], g["betas"][0], g["betas"][1], g["eps"], self._bc_dev, self.grad_scale)
        off = 0
        if self.overlap_from and ops.opt_stream_on() and a.flat.is_cuda and 0 < self.overlap_from < len(a.params):
            off = a.offsets[self.overlap_from]
        if off:
            # head (what the next forward needs first) here, tail on the optimizer stream: see ops.opt_stream
            ops.wait_late_params()                     # a previous tail nobody waited for (it reads the bias corrections)
            ops.adam_prepare(self._step_dev, self._bc_dev, g["betas"][0], g["betas"][1])
            ops.adam_apply(a.flat[:off], a.flat_grad[:off], self.exp_avg[:off], self.exp_avg_sq[:off], *hyper)
            cur = torch.cuda.current_stream(a.flat.device)
            st = ops.opt_stream(a.flat.device)
            st.wait_stream(cur)                        # gradients (and the all-reduce) are complete on this stream
            with torch.cuda.stream(st):
                ops.adam_apply(a.flat[off:], a.flat_grad[off:], self.exp_avg[off:], self.exp_avg_sq[off:], *hyper)
            a.late_start, a.late_event = off, st.record_event()
            ops.register_late(a)
        else:
            ops.adam_prepare(self._step_dev, self._bc_dev, g["betas"][0], g["betas"][1])
            ops.adam_apply(a.flat, a.flat_grad, self.exp_avg, self.exp_avg_sq, *hyper)
        a.epoch += 1
        ops.note_weights_changed()

    @property
    def step_count(self):
        """Blocking device-to-host read: not for use inside a captured step (graph.GraphedStep) or per parameter."""
        return int(self._step_dev.item())

    @step_count.setter
    def step_count(self, n):
        self._step_dev.fill_(int(n))

    def synchronize(self):
        """Orders the current stream after a pending optimizer tail (overlap_from): call before reading the parameters
        from code that does not go through hific_amd.Model (checkpointing, evaluation with another module, ...)."""
        ops.wait_late_params()

    def zero_grad(self, set_to_none=False):
        self.arena.zero_grad()

    # ---- checkpointing in torch.optim.Adam's own format (the reference saves `*_optimizer_state_dict`, -------------
    # ---- utils.py:131-137, and restores them in load_model, utils.py:191-197) ---------------------------------------
    def state_dict(self):
        """torch.optim.Adam's layout; `step` is a float32 0-d tensor (torch >= 1.12 convention; the reference's torch 1.6
        wrote a Python int - both load here)."""
        ops.wait_late_params()
        state = {}
        step_n = self.step_count               # ONE blocking device read (the count lives in device memory)
        if step_n > 0:
            for i, p in enumerate(self.arena.params):
                o, n = self.arena.slice_of(i)
                state[i] = dict(step=torch.tensor(float(step_n)),
                                exp_avg=self.exp_avg[o:o + n].view(p.shape).clone(),
                                exp_avg_sq=self.exp_avg_sq[o:o + n].view(p.shape).clone())
        g = self.param_groups[0]
        group = dict(lr=g["lr"], betas=tuple(g["betas"]), eps=g["eps"], weight_decay=0, amsgrad=False,
                     maximize=False, foreach=None, capturable=False, differentiable=False, fused=None,
                     params=list(range(len(self.arena.params))))
        return dict(state=state, param_groups=[group])

    def load_state_dict(self, sd):
        groups = sd["param_groups"]
        if len(groups) != 1 or len(groups[0]["params"]) != len(self.arena.params):
            raise ValueError("optimizer state_dict does not match this parameter group")
        if groups[0].get("weight_decay", 0) or groups[0].get("amsgrad", False):
            raise ValueError("FusedAdam implements torch.optim.Adam with weight_decay=0, amsgrad=False")
        g = self.param_groups[0]
        g["lr"], g["betas"], g["eps"] = groups[0]["lr"], tuple(groups[0]["betas"]), groups[0]["eps"]
        steps = set()
        ops.wait_late_params()             # a pending optimizer tail still reads / writes the moment buffers
        with torch.no_grad():
            self.exp_avg.zero_(); self.exp_avg_sq.zero_()
            for k, st in sd["state"].items():
                i = int(k)
                o, n = self.arena.slice_of(i)
                if st["exp_avg"].numel() != n or st["exp_avg_sq"].numel() != n:
                    raise ValueError(f"optimizer state of parameter {i} has {st['exp_avg'].numel()} elements, the "
                                     f"parameter {n}")
                self.exp_avg[o:o + n].copy_(st["exp_avg"].reshape(-1))
                self.exp_avg_sq[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
                steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError("FusedAdam keeps one step count per group; the state_dict holds several: %s" % sorted(steps))
        self.step_count = steps.pop() if steps else 0

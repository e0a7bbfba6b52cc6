"""Data-parallel training over RCCL/xGMI: one process per GPU, images sharded by rank, gradients summed with
bucketed all-reduces that overlap the remaining backward.

The reference has no multi-GPU path (train.py:303-308 raises NotImplementedError), so there is no call pattern to
mirror; the only semantics to keep is "gradient = mean over the global batch" (every loss is a batch mean).

Design for xGMI (7 point-to-point links per GPU, no switch): the gradient of a ParamArena is already one flat
float32 tensor, so a bucket is a contiguous slice (no flatten/copy); buckets are sealed in *reverse* arena order,
which is the order backward produces them, and each sealed bucket is handed to RCCL (`torch.distributed`, backend
"nccl" == RCCL on ROCm) as an async all-reduce on RCCL's own stream while the compute stream keeps running the
rest of backward.  Bucket size defaults to ~32 MiB: the 960x960x3x3 residual-block weights are 33 MB each, so a
bucket is about one such tensor — large enough for RCCL to spread over all links, small enough that the first
all-reduce starts after ~1/18 of the residual stack's backward.  The division by world size is folded into the
fused Adam kernel (grad_scale).
"""
import torch
import torch.distributed as dist


class BucketedGradReducer:
    def __init__(self, arena, bucket_mbytes=32, process_group=None, eager=True):
        self.arena = arena
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        cap = int(bucket_mbytes * (1 << 20) / 4)
        # contiguous slot ranges, built from the END of the arena (backward order)
        self.buckets = []          # (lo_elem, hi_elem, n_slots)
        self.slot_bucket = [0] * len(arena.slots)
        hi_slot = len(arena.slots)
        while hi_slot > 0:
            lo_slot = hi_slot - 1
            hi_elem = arena.numel if hi_slot == len(arena.slots) else arena.offsets[hi_slot]
            while lo_slot > 0 and hi_elem - arena.offsets[lo_slot - 1] <= cap:
                lo_slot -= 1
            b = len(self.buckets)
            for i in range(lo_slot, hi_slot):
                self.slot_bucket[i] = b
            self.buckets.append((arena.offsets[lo_slot], hi_elem, hi_slot - lo_slot))
            hi_slot = lo_slot
        self.pending = [0] * len(self.buckets)
        self.launched = [False] * len(self.buckets)
        self.works = []
        # eager=False: reduce everything in finish() (used for the Discriminator arena, whose slots are written by
        # two backward passes per optimizer step)
        self.active = self.world > 1 or (dist.is_initialized() and __import__('os').environ.get('HIFIC_FORCE_DIST') == '1')
        if self.active and eager:
            arena.on_write = self._on_write

    def _launch(self, b):
        lo, hi, _ = self.buckets[b]
        self.launched[b] = True
        self.works.append(dist.all_reduce(self.arena.flat_grad[lo:hi], op=dist.ReduceOp.SUM, group=self.pg,
                                          async_op=True))

    def _on_write(self, slot):
        b = self.slot_bucket[slot.index]
        self.pending[b] += 1
        if self.pending[b] == self.buckets[b][2] and not self.launched[b]:
            self._launch(b)

    def finish(self):
        """Launch whatever is not sealed yet (parameters that received no gradient this turn), wait for all
        collectives on the current stream, reset for the next backward.  Returns the gradient scale 1/world."""
        if self.active:
            for b in range(len(self.buckets)):
                if not self.launched[b]:
                    self._launch(b)
            for w in self.works:
                w.wait()
        self.works = []
        self.pending = [0] * len(self.buckets)
        self.launched = [False] * len(self.buckets)
        return 1.0 / self.world


def allreduce_scalar_mean(t, process_group=None):
    """Global-batch mean of a 0-d loss term (used for the rate-penalty branch so every rank picks the same lambda,
    SURVEY §8e)."""
    if dist.is_initialized() and dist.get_world_size(process_group) > 1:
        t = t.detach().clone()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=process_group)
        t /= dist.get_world_size(process_group)
    return t

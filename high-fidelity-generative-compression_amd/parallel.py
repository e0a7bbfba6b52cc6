"""Data-parallel training over RCCL/xGMI: one process per GPU, images sharded by rank, gradients summed with
bucketed all-reduces that overlap the remaining backward.

The reference has no multi-GPU path (train.py:303-308 raises NotImplementedError), so there is no call pattern to
mirror; the only semantics to keep is "gradient = mean over the global batch" (every loss is a batch mean) and one
rate-penalty branch for the whole job (`allreduce_scalar_mean`, used by loss/losses.py:weighted_rate_loss).

Design for xGMI (7 point-to-point links per GPU, no switch): the gradient of a ParamArena is already one flat
float32 tensor, so a bucket is a contiguous slice (no flatten/copy); buckets are sealed in *reverse* arena order,
which is the order backward produces them, and each sealed bucket is handed to RCCL (`torch.distributed`, backend
"nccl" == RCCL on ROCm) as an async all-reduce on RCCL's own stream while the compute stream keeps running the
rest of backward.  Bucket size defaults to 128 MiB ($HIFIC_BUCKET_MB): four of the 33 MB residual-block
weight tensors per collective, six collectives for the 726 MB of amortisation-model gradients - few, large transfers
for the per-link-bound xGMI ring, and the first one still starts after ~1/5 of the residual stack's backward (measured
with one rank, where the collective is a pure copy: 8 MiB 28.6 ms, 32 MiB 28.15, 128 MiB 27.8, 512 MiB 27.8 per cycle).
The slice sealed LAST (it holds the first layers of the model) is split again, [<= 2 MiB | <= 32 MiB | rest] from slot 0 upward
($HIFIC_BUCKET_TAIL_MB): what is on the wire when backward ends is exposed, so that piece is kept small.
Collectives are issued from a dedicated reduce stream that waits for the producing streams (ops.producer_streams), so
the backward pass never waits for its own weight gradients.  The division by world size is folded into the
fused Adam kernel (grad_scale).

Sealing rule: a bucket is sealed when every one of its slots has received its *expected number of writes* for this
backward (1 unless registered otherwise through `expected_writes`: a module applied twice per forward, such as
HyperpriorDensity when both of its likelihood evaluations are in the loss, writes its slots twice).  A write that
arrives after its bucket has been reduced cannot be repaired (the other ranks' contributions are already mixed in):
it raises instead of training on a silently wrong gradient.  Arenas whose write pattern is not fixed (the
Discriminator's slots are written by the G-turn and the D-turn of one optimizer step) use eager=False: everything is
reduced in finish().

Gradient payload (round 4): `payload="bf16"` ($HIFIC_GRAD_PAYLOAD=bf16) halves the bytes on the xGMI ring - a sealed bucket is
rounded to bfloat16 into a persistent staging buffer (on the reduce stream, hific_cast), the staging slice is all-reduced, and
finish() converts the sums back into the float32 gradient arena before Adam reads it, so the optimizer state and the
accumulation over steps stay float32.  Error per element: one rounding of each rank's term (2^-9 |g_r|) plus one rounding per
addition of the reduction; tests/test_ddp_gloo.py bounds the two-rank result by 2^-7 sum_r |g_r| element-wise.  The default
stays float32 (bit-identical to the sum of the shards).  DESIGN.md section 6 has the per-bucket time model.
"""
import os

import torch
import torch.distributed as dist


_NONBLOCK = os.environ.get("HIFIC_REDUCE_NONBLOCK", "1") not in ("0", "")


class BucketedGradReducer:
    def __init__(self, arena, bucket_mbytes=128, process_group=None, eager=True, expected_writes=None, payload=None,
                 tail_mbytes=None):
        """expected_writes: {parameter or slot index: writes per backward} for slots written more than once.
        payload: "f32" (default) or "bf16" (module docstring); $HIFIC_GRAD_PAYLOAD when None.
        tail_mbytes: caps (MiB) of the LAST-sealed buckets, from the start of the arena upward ($HIFIC_BUCKET_TAIL_MB, default
        "2,32"; eager arenas only).  Whatever is on the wire when backward ends is exposed: with plain 128 MiB slices that is the
        bucket holding the first layers of the model (67 MiB for the amortisation arena: the Encoder, the Generator's head and
        the first residual convolution - 0.37 ms on an 8-GPU ring).  Split as [first layers <= 2 MiB | <= 32 MiB | rest], the rest
        seals when the Generator's head is done, the 32 MiB piece in the middle of the Encoder's backward, and only ~1 MiB is
        left for the end."""
        self.arena = arena
        payload = payload or os.environ.get("HIFIC_GRAD_PAYLOAD", "f32")
        if payload not in ("f32", "bf16"):
            raise ValueError("gradient payload must be 'f32' or 'bf16'")
        self.payload = payload
        self.stage = None            # bf16 staging image of the gradient arena (allocated on first use)
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        bucket_mbytes = float(os.environ.get("HIFIC_BUCKET_MB", bucket_mbytes))
        cap = int(bucket_mbytes * (1 << 20) / 4)
        nslots = len(arena.slots)
        # contiguous slot ranges, built from the END of the arena (backward order)
        self.buckets = []          # (lo_elem, hi_elem, n_slots)
        self.slot_bucket = [0] * nslots
        hi_slot = nslots
        while hi_slot > 0:
            lo_slot = hi_slot - 1
            hi_elem = arena.numel if hi_slot == nslots else arena.offsets[hi_slot]
            while lo_slot > 0 and hi_elem - arena.offsets[lo_slot - 1] <= cap:
                lo_slot -= 1
            b = len(self.buckets)
            for i in range(lo_slot, hi_slot):
                self.slot_bucket[i] = b
            self.buckets.append((arena.offsets[lo_slot], hi_elem, hi_slot - lo_slot))
            hi_slot = lo_slot
        if tail_mbytes is None:
            tail_mbytes = [float(v) for v in os.environ.get("HIFIC_BUCKET_TAIL_MB", "2,32").split(",") if v.strip()]
        tail_caps = [int(float(v) * (1 << 20) / 4) for v in tail_mbytes if float(v) > 0]
        if eager and tail_caps and self.buckets:
            lo_e, hi_e, n_last = self.buckets.pop()             # the bucket that holds slot 0: sealed last
            b0 = len(self.buckets)
            ends = lambda i: hi_e if i == n_last else arena.offsets[i]      # element offset where slot i starts / the bucket ends
            pieces, start = [], 0
            for cap in tail_caps:
                end = start
                while end < n_last and ends(end + 1) - arena.offsets[start] <= cap:
                    end += 1
                if end == n_last:                               # everything left fits under this cap: no further split
                    break
                if end > start:
                    pieces.append((start, end))
                    start = end
            pieces.append((start, n_last))
            for lo_s, hi_s in reversed(pieces):                 # list order = backward order (end of the arena first)
                b = len(self.buckets)
                for i in range(lo_s, hi_s):
                    self.slot_bucket[i] = b
                self.buckets.append((arena.offsets[lo_s], ends(hi_s), hi_s - lo_s))
            assert self.buckets[-1][0] == lo_e and len(self.buckets) >= b0 + 1
        self.expected = [1] * nslots
        if expected_writes:
            index_of = {id(p): i for i, p in enumerate(arena.params)}
            for key, n in expected_writes.items():
                i = key if isinstance(key, int) else index_of[id(key)]
                self.expected[i] = int(n)
        self.works = []
        # exposed communication: GPU time the launching stream spends in finish() waiting for collectives that the backward
        # pass did not hide (event pairs, read by exposed_comm_ms() after a synchronize); off unless measure_exposed(True)
        # ... and, separately (measure_timeline), one event pair per bucket around its collective: that mode makes the reduce stream
        # wait for every collective before the next bucket's cast / collective is issued, i.e. it changes the schedule of the
        # reduce stream - never on during a timed headline
        self._measure, self._ev_pairs, self._timeline, self._bucket_ev, self._bucket_ev_last = False, [], False, [], []
        self.active = self.world > 1 or (dist.is_initialized() and os.environ.get('HIFIC_FORCE_DIST') == '1')
        self.eager = bool(eager)
        self._reset()
        if self.active and eager:
            arena.on_write = self._on_write

    def _reset(self):
        if self._bucket_ev:                                  # keep the LAST backward's events only (bounded)
            self._bucket_ev_last, self._bucket_ev = self._bucket_ev, []
        self.count = [0] * len(self.expected)
        self.remaining = [b[2] for b in self.buckets]      # slots of each bucket still short of their write count
        self.launched = [False] * len(self.buckets)

    def _to_stage(self, lo, hi):
        """float32 gradient slice -> bf16 staging slice (the tensor that goes on the wire); runs on the current stream."""
        if self.stage is None:
            self.stage = torch.empty(self.arena.numel, dtype=torch.bfloat16, device=self.arena.flat_grad.device)
        src, dst = self.arena.flat_grad[lo:hi], self.stage[lo:hi]
        if src.is_cuda:
            from . import lib
            lib.call("hific_cast", src.data_ptr(), lib.HIFIC_F32, dst.data_ptr(), lib.HIFIC_BF16, hi - lo, lib.stream())
        else:
            dst.copy_(src)
        return dst

    def _from_stage(self, lo, hi):
        src, dst = self.stage[lo:hi], self.arena.flat_grad[lo:hi]
        if src.is_cuda:
            from . import lib
            lib.call("hific_cast", src.data_ptr(), lib.HIFIC_BF16, dst.data_ptr(), lib.HIFIC_F32, hi - lo, lib.stream())
        else:
            dst.copy_(src)

    def _launch(self, b):
        lo, hi, _ = self.buckets[b]
        self.launched[b] = True
        grad = self.arena.flat_grad[lo:hi]
        from . import ops
        bf16 = self.payload == "bf16"
        if grad.is_cuda and _NONBLOCK:
            # The bucket's gradients were written by kernels on up to three streams (main, side = weight gradients, branch);
            # all of them are enqueued by now.  A dedicated reduce stream waits for those streams' current positions and the
            # collective is issued from it, so the backward pass itself never waits for its own weight gradients here
            # (ordering the main stream after the side stream at each of ~20 buckets cost 12 % of the step at one rank).
            dev = grad.device
            red = ops.reduce_stream(dev)
            for st in ops.producer_streams(dev):
                red.wait_stream(st)
            with torch.cuda.stream(red):
                wire = self._to_stage(lo, hi) if bf16 else grad
                if self._timeline:
                    # per-bucket timeline (bench.py `rccl.buckets_timeline`): issue = the reduce stream reaches the collective
                    # (every producer stream has delivered the bucket), done = the collective has finished.  The reduce stream
                    # is made to wait for the collective here, which changes nothing for the compute streams.
                    e0 = torch.cuda.Event(enable_timing=True); e0.record(red)
                    w = dist.all_reduce(wire, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
                    w.wait()
                    e1 = torch.cuda.Event(enable_timing=True); e1.record(red)
                    self._bucket_ev.append((b, (hi - lo) * (2 if bf16 else 4), e0, e1))
                    self.works.append(w)
                else:
                    self.works.append(dist.all_reduce(wire, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
            return
        # the collective is ordered after the CURRENT stream: bring the side / branch streams in
        if grad.is_cuda:
            ops.join_side_stream()
        wire = self._to_stage(lo, hi) if bf16 else grad
        self.works.append(dist.all_reduce(wire, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))

    def _on_write(self, slot):
        i = slot.index
        b = self.slot_bucket[i]
        if self.launched[b]:
            raise RuntimeError(
                f"gradient slot {i} ({tuple(self.arena.params[i].shape)}) was written after its bucket had been "
                f"all-reduced: it receives more than {self.expected[i]} write(s) per backward. Register the count with "
                f"BucketedGradReducer(..., expected_writes={{param: n}}) or use eager=False for this arena.")
        self.count[i] += 1
        if self.count[i] == self.expected[i]:
            self.remaining[b] -= 1
            if self.remaining[b] == 0:
                self._launch(b)

    def finish(self):
        """Launch whatever is not sealed yet (parameters that received fewer writes than expected this turn, or a
        deferred arena), wait for all collectives on the current stream, reset for the next backward.  Returns the
        gradient scale 1/world."""
        if self.active:
            for b in range(len(self.buckets)):
                if not self.launched[b]:
                    self._launch(b)
            timed = self._measure and self.arena.flat_grad.is_cuda
            if timed:
                cur = torch.cuda.current_stream(self.arena.flat_grad.device)
                e0 = torch.cuda.Event(enable_timing=True); e0.record(cur)
            for w in self.works:
                w.wait()
            if self.payload == "bf16":
                # the sums come home: bf16 staging image -> float32 gradient arena (ONE pass over the arena; Adam reads float32)
                self._from_stage(0, self.arena.numel)
            if timed:
                e1 = torch.cuda.Event(enable_timing=True); e1.record(cur)
                self._ev_pairs.append((e0, e1))
        self.works = []
        self._reset()
        return 1.0 / self.world


    def measure_exposed(self, on=True):
        """Event pair around the waits of finish() (exposed_comm_ms); does not change what runs where."""
        self._measure, self._ev_pairs = bool(on), []

    def measure_timeline(self, on=True):
        """Per-bucket issue / duration events (bucket_timeline).  Serialises the reduce stream behind every collective: use in a
        short separate pass, not while timing."""
        self._timeline, self._bucket_ev, self._bucket_ev_last = bool(on), [], []

    def bucket_timeline(self):
        """[(bucket, MiB on the wire, issue ms after the first bucket's issue, duration ms)] of the LAST backward run under
        measure_timeline(True); call after a device synchronize.  Empty when the collectives did not run on the reduce stream."""
        ev = self._bucket_ev or self._bucket_ev_last
        self._bucket_ev, self._bucket_ev_last = [], []
        if not ev:
            return []
        nb = len(self.buckets)
        last = ev[-nb:] if len(ev) >= nb else ev
        t0 = last[0][2]
        return [(b, round(nbytes / 2 ** 20, 2), round(t0.elapsed_time(e0), 3), round(e0.elapsed_time(e1), 3))
                for b, nbytes, e0, e1 in last]

    def exposed_comm_ms(self):
        """Total GPU time (ms) the launching stream waited inside finish() since measure_exposed(True); call after a
        device synchronize.  0.0 when nothing was measured."""
        ms = sum(a.elapsed_time(b) for a, b in self._ev_pairs)
        self._ev_pairs = []
        return ms


_SCALAR_COLLECTIVES = True


def set_scalar_collectives(on):
    """Measurement runs that step ONE rank of a job alone (bench.py scale_report: the other ranks are parked in a barrier) must
    not issue the model's own collective either: off = allreduce_scalar_mean is the identity.  Returns the previous setting."""
    global _SCALAR_COLLECTIVES
    was, _SCALAR_COLLECTIVES = _SCALAR_COLLECTIVES, bool(on)
    return was


def allreduce_scalar_mean(t, process_group=None):
    """Global-batch mean of a 0-d loss term (the rate-penalty branch: every rank must pick the same lambda,
    SURVEY section 8e).  Identity when torch.distributed is not initialised or the world has one rank."""
    if _SCALAR_COLLECTIVES and dist.is_initialized() and dist.get_world_size(process_group) > 1:
        t = t.detach().clone()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=process_group)
        t /= dist.get_world_size(process_group)
    return t

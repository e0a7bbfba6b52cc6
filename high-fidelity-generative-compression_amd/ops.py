"""torch.autograd.Function wrappers over the C-ABI (one Function per reference op call site).

Conventions
  * activations are contiguous NCHW, float32 (parity mode) or bfloat16 (fast mode); parameters are float32
  * `cd` = compute dtype code (lib.HIFIC_F32 / lib.HIFIC_BF16); in bf16 mode an op may read f32 inputs or
    write f32 outputs (flags) so the entropy-model chain stays f32 at the tensor level
  * nothing here computes with torch ops: torch allocates outputs and records the graph
"""
import ctypes
import os
import weakref

import torch
from torch.autograd import Function

from . import lib
from .lib import call, ptr, stream, workspace, require_gpu, HIFIC_F32, HIFIC_BF16

_compute_dtype = torch.float32


def set_compute_dtype(dt):
    """torch.float32 (parity, f32 MFMA) or torch.bfloat16 (bf16 MFMA, f32 accumulate)."""
    global _compute_dtype
    if dt not in (torch.float32, torch.bfloat16):
        raise ValueError("compute dtype must be torch.float32 or torch.bfloat16")
    _compute_dtype = dt


def get_compute_dtype():
    return _compute_dtype


# Bumped by every in-place parameter update that bypasses torch's version counters (optim.FusedAdam writes the
# arena with a raw kernel): cached derived data (packed bf16 weights) is keyed on it.
_weights_epoch = 0


def note_weights_changed():
    global _weights_epoch
    _weights_epoch += 1


def weights_epoch():
    return _weights_epoch


# ------------------------------------------------------------------------------------------------------
# Persistent packed-weight cache (include/hific_hip.h "persistent packed-weight cache").
#   key   = (weight storage address, direction, geometry, compute dtype), valid for ONE tensor object (weak reference)
#   token = (weight._version, epoch of the optim.ParamArena holding it) - what the packed image was made from.
#           Everything that updates parameters through torch (optimizers, load_state_dict, p.copy_/add_ ...) bumps
#           `_version`; optim.FusedAdam bumps the arena epoch.  NOT seen: in-place writes through an alias that has
#           its own version counter (`p.data.mul_(..)`): call ops.pack_cache.clear() after such edits, or run with
#           HIFIC_PACK_CACHE=0.
# A forward-type conv call whose entry is current passes wcache_state 2 and launches no pack kernel.  The first call
# that finds a stale entry re-packs EVERY stale entry in ONE launch (hific_pack_batch): after an optimizer step that is
# "all layers of that parameter group, both directions".  Inference never re-packs.
_PACK_CACHE_ON = __import__("os").environ.get("HIFIC_PACK_CACHE", "1") != "0"


class _PackEntry:
    __slots__ = ("weight", "buf", "bufptr", "bufbytes", "job", "token", "nblocks", "lds", "dtype", "kind", "event", "pack_sid",
                 "waited", "last_use", "pinned")


class WeightPackCache:
    def __init__(self):
        self.entries = {}
        self.prepared = {}          # tuple(entry keys) -> (jobs_dev, prefix_dev, total_blocks, lds, dtype)
        self.job_bytes = None
        self.tick = 0
        self.bytes = 0
        # Keys carry the full geometry, so inference over many image sizes would add a packed copy of every weight per
        # size: least-recently-used entries are dropped beyond this many bytes (one training configuration needs ~0.8 GB)
        self.cap_bytes = int(os.environ.get("HIFIC_PACK_CACHE_MB", "6144")) << 20
        # entries a captured hipGraph reads by raw pointer (graph.GraphedStep pins everything that exists at capture time:
        # replays never call lookup(), so their last_use would go stale and the LRU would free them first).  Dropping a
        # pinned entry (its weight died or moved) bumps pins_broken: every graph captured before that refuses to replay.
        self.pins_broken = 0

    def pin_all(self):
        """Pins every current entry on behalf of ONE graph (reference-counted: several captured graphs may share entries) and
        returns (pins_broken, [(key, entry)]); the graph hands the pairs back to unpin() when it is closed or collected.
        The pairs carry the entry OBJECTS: after clear() / a dead-entry sweep / an alias drop the same key can hold a fresh
        entry that a newer graph pinned - an old graph's unpin must not release that one."""
        pairs = list(self.entries.items())
        for _, e in pairs:
            e.pinned = int(e.pinned) + 1
        return self.pins_broken, pairs

    def unpin(self, pairs):
        for k, e in pairs:
            if self.entries.get(k) is e and e.pinned:
                e.pinned = int(e.pinned) - 1

    def _drop(self, key):
        e = self.entries.pop(key)
        self.bytes -= e.buf.numel()
        if e.pinned:
            self.pins_broken += 1
        # kernels on other streams may still read the packed image: the allocator must not recycle it under them
        for st in producer_streams(e.buf.device):
            e.buf.record_stream(st)

    def _make_room(self, device):
        if self.bytes <= self.cap_bytes:
            return
        for key, e in sorted(self.entries.items(), key=lambda kv: kv[1].last_use):
            if self.bytes <= self.cap_bytes // 2:
                break
            if not e.pinned:
                self._drop(key)
        self.prepared.clear()

    @staticmethod
    def _token(w):
        sl = getattr(w, "_hific_slot", None)
        return (w._version, sl.arena.epoch if sl is not None else -1)

    def clear(self):
        if any(e.pinned for e in self.entries.values()):
            self.pins_broken += 1
        self.entries.clear(); self.prepared.clear()
        self.bytes = 0

    def settle(self):
        """Call after a device synchronize: every pack issued so far is complete, so no stream needs to order itself after
        a pack event any more.  hific_amd.graph.GraphedStep does this before capturing - a stream inside a capture must not
        wait on an event recorded before the capture began."""
        for e in self.entries.values():
            e.event, e.pack_sid, e.waited = None, None, set()

    def lookup(self, weight, kind, geom, cd, flags, transposed):
        """-> (wcache ptr, bytes, state) for the C-ABI call."""
        if not _PACK_CACHE_ON:
            return None, 0, 0
        if self.job_bytes is None:
            self.job_bytes = int(lib.raw("hific_pack_job_bytes")())
        key = (weight.data_ptr(), kind, geom, cd, flags, weight.device.index)
        e = self.entries.get(key)
        tok = self._token(weight)
        if e is not None and e.weight() is not weight:
            # another tensor object at this address (the old one died, or an alias whose version counter we cannot
            # relate to the packed image): start over for this key (the old buffer leaves the byte count with it)
            self._drop(key)
            e = None
            self.prepared.clear()
        self.tick += 1
        if e is None:
            self._make_room(weight.device)
            e = _PackEntry()
            e.job = ctypes.create_string_buffer(self.job_bytes)
            fn = "hific_conv_transpose2d_pack_plan" if transposed else "hific_conv2d_pack_plan"
            call(fn, kind, *geom, cd, flags, e.job, self.job_bytes)
            nb, ld, wb, dt = ctypes.c_int(), ctypes.c_int(), ctypes.c_longlong(), ctypes.c_int()
            call("hific_pack_job_info", e.job, ctypes.byref(nb), ctypes.byref(ld), ctypes.byref(wb), ctypes.byref(dt))
            e.nblocks, e.lds, e.dtype = nb.value, ld.value, dt.value
            e.buf = torch.empty(int(wb.value), dtype=torch.uint8, device=weight.device)
            e.bufptr, e.bufbytes = e.buf.data_ptr(), e.buf.numel()
            e.weight = weakref.ref(weight)
            call("hific_pack_job_set_ptrs", e.job, e.buf.data_ptr(), weight.data_ptr(), None)
            e.token = None                       # never packed yet
            e.kind, e.event, e.pack_sid, e.waited, e.pinned = kind, None, None, set(), False
            self.entries[key] = e
            self.bytes += e.buf.numel()
            self.prepared.clear()
        e.last_use = self.tick
        if e.token != tok:
            self.refresh_stale()
        # packed on another stream (the batched re-pack runs on the stream of the first stale lookup; data-gradient packs on
        # a dedicated one): order this stream after it, once per re-pack
        if e.event is not None:
            sid = stream()
            if sid != e.pack_sid and sid not in e.waited:
                lib.stream_obj().wait_event(e.event)
                e.waited.add(sid)
        return e.bufptr, e.bufbytes, 2

    def refresh_stale(self):
        """Re-packs every entry whose weight changed since it was packed (or that was never packed), one batched launch per
        (dtype, direction).  Forward packs (kind 0) run on the current stream - the caller needs one of them right now;
        data-gradient packs (kind 1) are first needed in the next backward pass and run on a separate stream, off the
        critical path when HIFIC_PACK_STREAM=1 (default 0: everything on the current stream - measured faster)."""
        split_weights.refresh()        # derived (hi, hi, lo) weight images first: their packed images are entries here
        # dead: the weight object is gone, or it no longer lives at the address this entry was keyed (and its pack job was
        # pointed) at - `model.cpu().to(dev)` before ParamArena.rebind(): re-packing from the old address would read freed
        # memory, and nothing would ever look the entry up again
        dead = [k for k, e in self.entries.items() if e.weight() is None or e.weight().data_ptr() != k[0]]
        for k in dead:
            self._drop(k)
        if dead:
            self.prepared.clear()
        stale = [(k, e) for k, e in self.entries.items() if e.token != self._token(e.weight())]
        if not stale:
            return
        dev = stale[0][1].buf.device
        cur = torch.cuda.current_stream(dev)
        late = {k for k, e in stale if _is_late(e.weight())}
        for dt in (HIFIC_BF16, HIFIC_F32):
            # (late?, kinds): weights whose optimizer update is still running on the optimizer stream are packed there,
            # right behind it - forward packs first, they are needed first
            plans = [(False, (0,)), (False, (1,))] if _PACK_STREAM_ON else [(False, (0, 1))]
            if late:
                plans += [(True, (0,)), (True, (1,))]
            for is_late, kinds in plans:
                group = [(k, e) for k, e in stale if e.dtype == dt and e.kind in kinds and ((k in late) == is_late)]
                if not group:
                    continue
                sig = tuple(k for k, _ in group)
                prep = self.prepared.get(sig)
                if prep is None:
                    raw = b"".join(bytes(e.job.raw) for _, e in group)
                    jobs_dev = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
                    pref, tot = [], 0
                    for _, e in group:
                        pref.append(tot); tot += e.nblocks
                    prefix_dev = torch.tensor(pref, dtype=torch.int32, device=dev)
                    prep = (jobs_dev, prefix_dev, tot, max(e.lds for _, e in group))
                    self.prepared[sig] = prep
                jobs_dev, prefix_dev, tot, lds_b = prep
                st = cur
                if is_late:
                    st = opt_stream(dev)             # FIFO behind the optimizer tail that writes these weights
                    jobs_dev.record_stream(st); prefix_dev.record_stream(st)
                elif _PACK_STREAM_ON and kinds == (1,):
                    st = _pack_stream(dev)
                    st.wait_stream(cur)              # the weights were written (optimizer) on / before the current stream
                    jobs_dev.record_stream(st); prefix_dev.record_stream(st)
                call("hific_pack_batch", jobs_dev.data_ptr(), prefix_dev.data_ptr(), len(group), tot, lds_b, dt, st.cuda_stream)
                ev = st.record_event()
                for _, e in group:
                    e.token = self._token(e.weight())
                    e.event, e.pack_sid, e.waited = ev, st.cuda_stream, set()


# ---- optimizer tail on its own stream -----------------------------------------------------------------------------------
# FusedAdam(overlap_from=k) updates the first k parameters (the Encoder: what the next forward pass needs first) on the
# current stream and the rest - plus their weight re-packs, see refresh_stale - on this stream, so the next step's Encoder
# forward overlaps with ~70 % of the (HBM-bound) optimizer + pack work.  The model orders itself after the tail with
# wait_late_params() before it touches those parameters (hific_amd.Model: right after the Encoder).
# Measured (round 2, batch 16 x 256^2): 26.0 -> 26.4 ms per GAN cycle, 16.8 -> 17.3 ms per compression step - the HBM-bound
# optimizer / pack kernels slow the Encoder convolutions they run next to by more than they hide.  Opt-in.
_OPT_STREAM_ON = os.environ.get("HIFIC_OPT_STREAM", "0") not in ("0", "")
_opt_streams = {}
_late_arenas = []


def opt_stream_on():
    return _OPT_STREAM_ON


def opt_stream(device):
    idx = device.index if device.index is not None else torch.cuda.current_device()
    st = _opt_streams.get(idx)
    if st is None:
        st = _opt_streams[idx] = torch.cuda.Stream(device=device)
    return st


def register_late(arena):
    if arena not in _late_arenas:
        _late_arenas.append(arena)


def wait_late_params():
    """Orders the current stream after every pending optimizer tail (no-op when there is none)."""
    while _late_arenas:
        a = _late_arenas.pop()
        if a.late_event is not None:
            torch.cuda.current_stream(a.flat.device).wait_event(a.late_event)
            a.late_event = None


def _is_late(weight):
    sl = getattr(weight, "_hific_slot", None)
    if sl is None:
        return False
    a = sl.arena
    return getattr(a, "late_event", None) is not None and a.offsets[sl.index] >= a.late_start


# measured (round 2): the HBM-bound re-pack on its own stream slows the forward pass it overlaps with more than it saves
# (26.8 -> 27.2 ms per GAN cycle): opt-in
_PACK_STREAM_ON = os.environ.get("HIFIC_PACK_STREAM", "0") not in ("0", "")
_pack_streams = {}


def _pack_stream(device):
    idx = device.index if device.index is not None else torch.cuda.current_device()
    st = _pack_streams.get(idx)
    if st is None:
        st = _pack_streams[idx] = torch.cuda.Stream(device=device, priority=_SIDE_PRIO)
    return st


pack_cache = WeightPackCache()


# ------------------------------------------------------------------------------------------------------
# Exact-index mode.  The reference floors y - mu + 0.5 into the latent indices
# (src/hyperprior.py:68-74,108-122); with bf16 operands in the chain that produces y and mu (Encoder -> analysis ->
# synthesis_mu) 0.3-0.4 % of the indices flip by one.  In this mode the FORWARD contractions of that chain run with
# split-bf16 operands (hific_split3: x*w ~= xh*wh + xl*wh + xh*wl over 3C reduction channels of the ordinary bf16 MFMA
# kernels, f32 accumulate, f32 activations in between); everything else - and the whole backward pass - stays bf16.
# Only meaningful in bf16 compute mode (float32 mode is exact already).  HIFIC_EXACT_INDEX=0 restores the plain bf16 chain.
_EXACT_INDEX = os.environ.get("HIFIC_EXACT_INDEX", "1") not in ("0", "")


def set_exact_index(on):
    global _EXACT_INDEX
    _EXACT_INDEX = bool(on)


def exact_index_on():
    return _EXACT_INDEX


# Exact-reconstruction option: north_star asks for reconstructions within 1e-3 of the reference;
# with bf16 activations through 24 Generator convolutions + 25 ChannelNorms the reconstruction given equal indices is 1.3e-2
# off.  With this option every NO-GRAD Generator forward (Model.decompress, the EVALUATION forward: src/model.py:312-344,
# 357-366) runs its contractions with split-bf16 operands and keeps float32 activations, like the exact-index chain.
# Off by default (the forward costs ~2x); HIFIC_EXACT_RECON=1 / hific_amd.set_exact_reconstruction(True).
_EXACT_RECON = os.environ.get("HIFIC_EXACT_RECON", "0") not in ("0", "")


def set_exact_reconstruction(on):
    global _EXACT_RECON
    _EXACT_RECON = bool(on)


def exact_reconstruction_on():
    return _EXACT_RECON


# Exact-TRAINING option: the same split-bf16 Generator forward WITH autograd, so that a training step
# - not only decompress / evaluation - produces a reconstruction within north_star's 1e-3 of the reference's.  The Generator then
# keeps float32 activations (Conv2dFn / ConvTranspose2dFn exact=True: split-bf16 forward, float32 output), its backward runs the
# bf16 MFMA kernels on those float32 tensors (operands rounded to bf16 in the tile loaders, f32 accumulate) - the arithmetic of
# the hyper nets' exact chain.  Costs the trunk three MFMAs per product and float32 activation traffic: the bench reports it
# as `exact_training`; the default training mode stays plain bf16 in the Generator.  HIFIC_EXACT_TRAIN=1 /
# hific_amd.set_exact_training(True).
_EXACT_TRAIN = os.environ.get("HIFIC_EXACT_TRAIN", "0") not in ("0", "")


def set_exact_training(on):
    global _EXACT_TRAIN
    _EXACT_TRAIN = bool(on)


def exact_training_on():
    return _EXACT_TRAIN


# The Generator under the exact-training / exact-reconstruction options as ONE chain of fused conv -> norm blocks
# (network/generator.py::_forward_exact_chain; round 6): float32-accurate forward values, bf16 stored activations, the plain
# bf16 backward pass.  HIFIC_EXACT_GEN_FUSED=0: the round-5 form (float32 activations between separate conv / norm ops).
_EXACT_GEN_FUSED = os.environ.get("HIFIC_EXACT_GEN_FUSED", "1") not in ("0", "")


def exact_generator_fused_on():
    return _EXACT_GEN_FUSED


def set_exact_generator_fused(on):
    global _EXACT_GEN_FUSED
    _EXACT_GEN_FUSED = bool(on)


# conv -> ChannelNorm pairs of the plain path as one autograd node (ConvNormFn; host time only, bit-identical): HIFIC_CONV_NORM_FUSED=0
# restores one node per op
_CONV_NORM_FUSED = os.environ.get("HIFIC_CONV_NORM_FUSED", "1") not in ("0", "")


def conv_norm_fused_on():
    return _CONV_NORM_FUSED


def set_conv_norm_fused(on):
    global _CONV_NORM_FUSED
    _CONV_NORM_FUSED = bool(on)


class exact_index_suspended:
    """`with ops.exact_index_suspended():` - the plain bf16 chain for the calls inside (restores the previous setting)."""

    def __enter__(self):
        global _EXACT_INDEX
        self._was = _EXACT_INDEX
        _EXACT_INDEX = False
        return self

    def __exit__(self, *exc):
        global _EXACT_INDEX
        _EXACT_INDEX = self._was
        return False


# The split images of the exact chain carry 3C channels per image and are addressed with 32-bit element offsets
# (hific_split3 / hific_channelnorm_fwd_exact return HIFIC_ERR_UNSUPPORTED from 3*C*H*W >= 2^31 on): a 60-channel plane
# reaches that at ~11.9 megapixels.  The plain bf16 chain addresses C*H*W and keeps working there.
_EXACT_MAX_ELEMS = 1 << 31


def exact_chain_fits(planes):
    """planes: iterable of (channels, H, W) of every tensor the exact chain would hold a split image of."""
    return all(3 * c * h * w < _EXACT_MAX_ELEMS for c, h, w in planes)


# Encoder blocks of the exact chain as one op each (ExactConvNormFn: bf16 autograd graph, split images handed from norm to
# conv); HIFIC_EXACT_FUSED=0: the round-3 first version (float32 activations + hific_split3 passes, float32 backward)
_EXACT_FUSED = os.environ.get("HIFIC_EXACT_FUSED", "1") not in ("0", "")


def fused_exact_blocks_on():
    return _EXACT_FUSED


class _SplitEntry:
    __slots__ = ("weight", "w3", "token", "addr", "transposed", "layout")


# Operand layouts of the exact chain.  SPLIT_3C: (hi, lo, hi) x (hi, hi, lo) over 3C reduction channels of the ordinary bf16
# kernels (round 3).  SPLIT_PAIR: both operands as (hi 16 | lo 16) groups over 2 * C16 channels, cross terms formed by the
# native split kernel (gconv_kernel SPLIT; hific_split3 which = 2): a third less staging, LDS traffic and barrier steps for
# the same three MFMAs.  HIFIC_EXACT_PAIR=0 keeps every layer on the 3C form.
SPLIT_3C, SPLIT_PAIR = 0, 2
_EXACT_PAIR = os.environ.get("HIFIC_EXACT_PAIR", "1") not in ("0", "")


def exact_pair_on():
    return _EXACT_PAIR


def set_exact_pair(on):
    global _EXACT_PAIR
    _EXACT_PAIR = bool(on)


def pair_channels(C):
    return 2 * ((C + 15) // 16 * 16)


class SplitWeightCache:
    """(hi, hi, lo) images of the weights of the exact-index chain along their reduction-channel dimension: float32 tensors
    [K, 3C, R, S] (conv) / [3Ci, Co, R, S] (conv-transpose) whose values are exactly bf16-representable, so the ordinary
    weight-pack path turns them into the packed bf16 operand.  Refreshed (all stale entries together, before the batched
    re-pack looks at them) when the source weight's (version, arena epoch) token changes."""

    def __init__(self):
        self.entries = {}

    def clear(self):
        self.entries.clear()

    def get(self, weight, transposed, layout=SPLIT_3C):
        key = (id(weight), layout)
        e = self.entries.get(key)
        if e is not None and e.weight() is not weight:
            e = None
        if e is None:
            e = _SplitEntry()
            e.weight, e.transposed, e.token, e.addr, e.w3 = weakref.ref(weight), transposed, None, None, None
            e.layout = layout
            self.entries[key] = e
        if e.token != WeightPackCache._token(weight) or e.addr != weight.data_ptr():
            self.refresh()
        return e.w3

    def refresh(self):
        dead = [k for k, e in self.entries.items() if e.weight() is None]
        for k in dead:
            del self.entries[k]
        waited = False
        for e in self.entries.values():
            w = e.weight()
            tok = WeightPackCache._token(w)
            if e.token == tok and e.addr == w.data_ptr():
                continue
            if not w.is_cuda:
                continue
            if not waited and _is_late(w):
                wait_late_params(); waited = True
            shape = list(w.shape)
            cdim = 0 if e.transposed else 1
            if e.transposed:
                outer, C, inner = 1, shape[0], w.numel() // shape[0]
            else:
                outer, C, inner = shape[0], shape[1], shape[2] * shape[3]
            shape[cdim] = pair_channels(C) if e.layout == SPLIT_PAIR else 3 * C
            if e.w3 is None or e.w3.device != w.device or list(e.w3.shape) != shape:
                e.w3 = torch.empty(shape, dtype=torch.float32, device=w.device)
            with torch.cuda.device(w.device):
                call("hific_split3", w.data_ptr(), e.w3.data_ptr(), outer, C, inner, 2 if e.layout == SPLIT_PAIR else 1,
                     HIFIC_F32, torch.cuda.current_stream(w.device).cuda_stream)
            torch.autograd.graph.increment_version(e.w3)        # the pack cache keys on it
            e.token, e.addr = tok, w.data_ptr()


split_weights = SplitWeightCache()


def _split3_act(x, layout=SPLIT_3C):
    """float32 [N, C, H, W] -> bf16 [N, 3C, H, W] = (hi, lo, hi), or the pair layout [N, 2 * C16, H, W]."""
    if x.dtype != torch.float32:
        raise lib.HificError("exact-index convolutions take float32 activations")
    N, C, H, W = x.shape
    Cx = pair_channels(C) if layout == SPLIT_PAIR else 3 * C
    x3 = torch.empty((N, Cx, H, W), dtype=torch.bfloat16, device=x.device)
    call("hific_split3", ptr(x), ptr(x3), N, C, H * W, 2 if layout == SPLIT_PAIR else 0, HIFIC_BF16, stream())
    return x3


def _exact_flags(layout, C):
    """conv forward flags of an exact layer: float32 output, + 3C marker (profiler) or pair layout + real channel count."""
    return (1 << 1) | ((1 << 3) | (C << 8) if layout == SPLIT_PAIR else (1 << 2))


# split-in-pack (hific_conv2d_fwd flags bit 5, round 6): for a 3C-layout exact nn.Conv2d whose channel count is a multiple of 64
# (the residual trunk, the 960 -> 220 Encoder layer) the packed (hi, hi, lo) operand is formed by the weight-pack pass from the
# float32 master weight itself - no derived [K, 3C, R, S] float32 image (SplitWeightCache) is written and re-read after every
# optimizer step (34 -> 10 bytes per weight and step).  HIFIC_SPLIT_IN_PACK=0 restores the derived image.
_SPLIT_IN_PACK = os.environ.get("HIFIC_SPLIT_IN_PACK", "1") not in ("0", "")


def _split_in_pack(weight, layout, transposed=False):
    return _SPLIT_IN_PACK and layout == SPLIT_3C and not transposed and weight.shape[1] % 64 == 0 and _PACK_CACHE_ON


def set_split_in_pack(on):
    global _SPLIT_IN_PACK
    _SPLIT_IN_PACK = bool(on)


def _wcache(weight, kind, geom, cd, flags, w_scale=None, transposed=False):
    """Cache arguments of a forward-type conv call; spectral-norm convs (w_scale changes every forward) bypass it."""
    if w_scale is not None or not weight.is_cuda:
        return None, 0, 0
    return pack_cache.lookup(weight, kind, geom, cd, flags, transposed)


def _cd():
    return HIFIC_F32 if _compute_dtype == torch.float32 else HIFIC_BF16


def _is_f32(t):
    return 1 if t.dtype == torch.float32 else 0


def _ws(t):
    w = workspace(t.device)
    return w.data_ptr(), w.numel()


def _slot(t):
    """Gradient slot bound by optim.ParamArena (None -> gradients are returned to autograd as tensors)."""
    return getattr(t, "_hific_slot", None) if t is not None else None


def _grad_target(slot, like):
    """(tensor to write, accumulate flag, value to return to autograd)."""
    if slot is None:
        t = torch.empty_like(like)
        return t, 0, t
    return slot.grad, slot.take(), None


def _written(*slots):
    for sl in slots:
        if sl is not None:
            sl.written()


# ------------------------------------------------------------------------------------------------------
# Weight / bias gradients on a side stream.  A layer's weight gradient depends only on (x, dy) and nothing on the rest of
# the backward pass depends on it, so it is launched on a second HIP stream and runs concurrently with the data-gradient
# chain: the tail of one kernel's grid (e.g. 746 workgroups on 512 slots) is filled by the other's workgroups.  Only for
# parameters that live in a ParamArena (the kernel writes the arena's gradient slot; nothing is handed back to autograd,
# which would read it on the main stream).  Joined (main waits for side) by an autograd-engine callback at the end of
# every backward pass and by `join_side_stream()` (the DDP reducer calls it before a bucket's all-reduce).
_SIDE_ON = os.environ.get("HIFIC_SIDE_WGRAD", "1") not in ("0", "")
# HIFIC_SIDE_STREAMS side streams per device, one per gradient slot (hashed), each with its own workspace.  Default 1: with one,
# the weight gradients of the Discriminator's layers and their reduce / finalize / spectral-norm chains queue behind each other
# for ~0.2 ms after the main stream has finished the D-turn's data-gradient chain (tools/r05/tail.py), but two or three streams
# measured the same cycle time within noise (1572 / 1570 / 1574 images/s, three runs each: round 5)
_N_SIDE = max(1, int(os.environ.get("HIFIC_SIDE_STREAMS", "1")))
# stream priority of the weight-gradient (side) and pack streams (HIP: lower number = higher priority; 0 = default): a positive
# value lets the critical-path kernels of the main stream win the CUs (experiment knob, round 6)
_SIDE_PRIO = int(os.environ.get("HIFIC_SIDE_PRIO", "0"))
_side_streams = {}            # device index -> [streams]
_side_state = {"pending": False, "cb": False, "rr": 0}


def set_side_stream(on):
    global _SIDE_ON
    _SIDE_ON = bool(on)


def _stream_capturing(st):
    with torch.cuda.stream(st):
        return torch.cuda.is_current_stream_capturing()


def _may_wait(cur, other):
    """Inside a hipGraph capture a stream may only wait for streams that are part of the same capture; one that took no work
    in this capture has nothing to wait for anyway."""
    return not torch.cuda.is_current_stream_capturing() or _stream_capturing(other)


def join_side_stream():
    """Make the current stream wait for every weight-gradient kernel launched on the side stream so far."""
    if _side_state["pending"] or _branch_streams:
        for dev_index, sides in _side_streams.items():
            cur = torch.cuda.current_stream(dev_index)
            for side in sides:
                if _may_wait(cur, side):
                    cur.wait_stream(side)
        # kernels of ops whose forward ran on the branch stream write parameter gradients (arena slots) from that stream in
        # backward; autograd's own end-of-backward synchronisation only covers gradients it accumulates itself
        for dev_index, br in _branch_streams.items():
            cur = torch.cuda.current_stream(dev_index)
            if not _may_wait(cur, br):
                continue
            if cur.cuda_stream != br.cuda_stream:
                cur.wait_stream(br)
            else:
                # called from a backward node that itself runs on the branch stream (a DDP bucket sealed there): the
                # collective launched next is ordered after THIS stream, so it must also see the caller's main stream
                home = _home_streams.get(dev_index)
                if home is not None:
                    cur.wait_stream(home)
        _side_state["pending"] = False


_reduce_streams = {}


def reduce_stream(device):
    """Stream the DDP reducer issues its collectives from (parallel.BucketedGradReducer)."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    st = _reduce_streams.get(idx)
    if st is None:
        st = _reduce_streams[idx] = torch.cuda.Stream(device=device)
    return st


def producer_streams(device):
    """Every stream gradient kernels may have been enqueued on: the current one, the model's main stream, side, branch."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    out = [torch.cuda.current_stream(idx)]
    cands = [_home_streams.get(idx)] + list(_side_streams.get(idx, ())) + [_branch_streams.get(idx)]
    for st in cands:
        if st is not None and all(st.cuda_stream != o.cuda_stream for o in out):
            out.append(st)
    return out


def _join_callback():
    _side_state["cb"] = False
    join_side_stream()


class _SideLaunch:
    """`with _SideLaunch(ev, x, dy):` - launches inside run on the side stream after event `ev` of the main stream."""

    def __init__(self, ev, *tensors, key=None):
        """`key`: the gradient slot the launches write (its arena index picks the side stream, so that two uses of one
        parameter in a backward pass - overwrite, then accumulate - stay ordered); None: round-robin."""
        dev = tensors[0].device
        sides = _side_streams.get(dev.index)
        if sides is None:
            sides = _side_streams[dev.index] = [torch.cuda.Stream(device=dev, priority=_SIDE_PRIO) for _ in range(_N_SIDE)]
        if key is not None:
            side = sides[((int(key.index) * 2654435761) >> 16) % len(sides)]      # (weights sit at every other index)
        else:
            side = sides[_side_state["rr"] % len(sides)]
            _side_state["rr"] += 1
        side.wait_event(ev)
        for t in tensors:
            if t is not None:
                t.record_stream(side)           # the caching allocator must not recycle them under the side kernel
        self._side = side

    # (torch.cuda.stream()'s context manager re-queries the current stream and device on both ends: ~20 us per use, ~60 uses
    # per training cycle; the side stream lives on the tensors' device, which is the current one inside an op)
    def __enter__(self):
        self._prev = lib.stream_obj()
        sd = self._side
        torch._C._cuda_setStream(stream_id=sd.stream_id, device_index=sd.device_index, device_type=sd.device_type)
        return self

    def __exit__(self, *exc):
        pv = self._prev
        torch._C._cuda_setStream(stream_id=pv.stream_id, device_index=pv.device_index, device_type=pv.device_type)
        _side_state["pending"] = True
        if not _side_state["cb"]:
            _side_state["cb"] = True
            torch.autograd.Variable._execution_engine.queue_callback(_join_callback)
        return False


# A second "branch" stream for independent sub-graphs of the forward pass (model.py: loss branch vs Discriminator branch)
_BRANCH_ON = os.environ.get("HIFIC_BRANCH_STREAMS", "1") not in ("0", "")
_branch_streams = {}


def branch_streams_on():
    return _BRANCH_ON


# which of the three branch-stream uses are active (bit 0: LPIPS target prefetch, bit 1: hyperprior rate side, bit 2: loss
# branch next to the Discriminator branch); HIFIC_BRANCH_MASK for experiments
_BRANCH_MASK = int(os.environ.get("HIFIC_BRANCH_MASK", "7"))


def branch_use(bit):
    return _BRANCH_ON and bool(_BRANCH_MASK & (1 << bit))


def set_branch_mask(mask):
    global _BRANCH_MASK
    _BRANCH_MASK = int(mask)


def set_branch_streams(on):
    global _BRANCH_ON
    _BRANCH_ON = bool(on)


_home_streams = {}


def branch_stream(device):
    """The branch stream of `device`; the stream that is current when it is requested is remembered as the model's main
    stream (callers ask for it right before forking work off their current stream)."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    st = _branch_streams.get(idx)
    if st is None:
        st = _branch_streams[idx] = torch.cuda.Stream(device=device)
    cur = torch.cuda.current_stream(idx)
    if cur.cuda_stream != st.cuda_stream:
        _home_streams[idx] = cur
    return st


def _use_side(*slots):
    return _SIDE_ON and all(sl is not None for sl in slots)


def _c16(t):
    """Contiguous AND 16-byte aligned: the pipelined / weight-resident conv kernels move 16-byte pieces and their launch plans
    are cached without pointers, so a contiguous view at an odd storage offset (a narrowed / as_strided tensor) is copied
    here instead of failing inside the library.  torch's own allocations are 512-byte aligned: no copy on the
    training path."""
    t = t.contiguous()
    return t if (t.data_ptr() & 15) == 0 else t.clone(memory_format=torch.contiguous_format)


def _act_code(act):
    return {None: lib.ACT_NONE, "none": lib.ACT_NONE, "relu": lib.ACT_RELU, "leaky_relu": lib.ACT_LEAKY}[act]


# ------------------------------------------------------------------------------------------------------
class Conv2dFn(Function):
    """y = act(conv2d(pad(x)) + b).  geom = (stride, pt, pl, pb, pr, pad_mode)."""

    @staticmethod
    def forward(ctx, x, weight, bias, geom, act, out_f32, w_scale, exact=False, bias_grad=True, x3=None):
        require_gpu(x, weight, bias)
        cd = _cd()
        ctx.bias_grad = bool(bias_grad)      # False: the ChannelNorm behind this layer produces it (ChannelNormFn prev_bias)
        stride, pt, pl, pb, pr, pad_mode = geom
        N, C, H, W = x.shape
        K, Cw, R, S = weight.shape
        assert Cw == C, "channel mismatch"
        OH = (H + pt + pb - R) // stride + 1
        OW = (W + pl + pr - S) // stride + 1
        # exact: False | True | "pair" (exact-index chain) | "recon" | "recon_pair" (a Generator layer under the exact-training /
        # exact-reconstruction options, network/layers.py::_exact_mode - only THOSE take the bf16 backward below)
        recon = exact in ("recon", "recon_pair")
        lay = SPLIT_PAIR if (exact in ("pair", "recon_pair") and C >= 16) else SPLIT_3C
        exact = bool(exact) and cd == HIFIC_BF16 and w_scale is None
        ydt = torch.float32 if (cd == HIFIC_F32 or out_f32 or exact) else torch.bfloat16
        if cd == HIFIC_F32 and x.dtype != torch.float32:
            raise lib.HificError("float32 compute mode needs float32 activations")
        y = torch.empty((N, K, OH, OW), dtype=ydt, device=x.device)
        wsp, wsb = _ws(x)
        if exact:
            # split-bf16 forward: 3C reduction channels of (hi, lo, hi) x (hi, hi, lo) (flags bit2 = count C, not 3C, FLOPs), or
            # exact="pair": both operands in the pair layout, cross terms formed by the native split kernel (flags bit3)
            # x3 given (ExactConvNormFn chain): x is the nominal bf16 activation, only saved for the weight gradient
            x3 = x3 if x3 is not None else _split3_act(x, lay)
            Cx = x3.shape[1]
            # a caller-supplied split image must be in the layout the weights are taken in (a pair-layout activation against
            # 3C weights - or the reverse - would contract silently to garbage)
            if Cx != (pair_channels(C) if lay == SPLIT_PAIR else 3 * C):
                raise lib.HificError(f"conv2d(exact): split image has {Cx} channels, layout {lay} of {C} channels needs "
                                     f"{pair_channels(C) if lay == SPLIT_PAIR else 3 * C}")
            flags = _exact_flags(lay, C)
            if _split_in_pack(weight, lay):
                w3, flags = weight, flags | 32               # the pack pass forms (hi, hi, lo) from the master weight
            else:
                w3 = split_weights.get(weight, transposed=False, layout=lay)
            wc = _wcache(w3, 0, (N, Cx, H, W, K, R, S, stride, pt, pl, pb, pr, pad_mode), cd, flags & 0xff, None)
            call("hific_conv2d_fwd", ptr(x3), ptr(w3), None, ptr(bias), None, ptr(y),
                 N, Cx, H, W, K, R, S, stride, pt, pl, pb, pr, pad_mode, _act_code(act), cd, flags, wsp, wsb, *wc, stream())
        else:
            flags = (_is_f32(x) if cd == HIFIC_BF16 else 0) | ((1 if ydt == torch.float32 else 0) << 1 if cd == HIFIC_BF16 else 0)
            wc = _wcache(weight, 0, (N, C, H, W, K, R, S, stride, pt, pl, pb, pr, pad_mode), cd, flags, w_scale)
            call("hific_conv2d_fwd", ptr(x), ptr(weight), ptr(w_scale), ptr(bias), None, ptr(y),
                 N, C, H, W, K, R, S, stride, pt, pl, pb, pr, pad_mode, _act_code(act), cd, flags, wsp, wsb, *wc, stream())
        ctx.geom = geom
        ctx.act = act
        ctx.cd = cd
        ctx.has_bias = bias is not None
        ctx.w_slot, ctx.b_slot = _slot(weight), _slot(bias)
        # exact-training option: the backward of an exact layer is the plain bf16 one - the tile loaders would round the float32
        # activation / gradient to bf16 anyway, and only bf16 operands reach the fast kernels (wgrad_s1 / wgrad_s2 / gconv_sp9 RFX;
        # with float32 operands the residual-block weight gradient ran 317 us on the generic kernel against 88 us).  Saves the
        # bf16 image of x instead of x.
        ctx.bf16_bwd = bool(exact and recon and torch.is_grad_enabled() and x.dtype == torch.float32)
        ctx.x_dtype = x.dtype
        xs = cast(x, torch.bfloat16) if ctx.bf16_bwd else x
        ctx.save_for_backward(xs, weight, w_scale, y if act not in (None, "none") else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, w_scale, y = ctx.saved_tensors
        stride, pt, pl, pb, pr, pad_mode = ctx.geom
        cd = ctx.cd
        N, C, H, W = x.shape
        K, _, R, S = weight.shape
        dy = _c16(dy)
        wsp, wsb = _ws(x)
        if y is not None:
            dz = torch.empty_like(dy)
            slope = 0.0 if ctx.act == "relu" else 0.2
            call("hific_act_bwd", ptr(dy), ptr(y), ptr(dz), dy.numel(), slope, lib.dtype_code(dy), stream())
            dy = dz
        if ctx.bf16_bwd and dy.dtype != torch.bfloat16:
            dy = cast(dy, torch.bfloat16)
        dy_f32 = _is_f32(dy) if cd == HIFIC_BF16 else 0
        dx = dw = db = None
        want_w, want_b = ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2] and ctx.bias_grad
        side = (want_w or want_b) and _use_side(ctx.w_slot if want_w else True, ctx.b_slot if want_b else True)
        ev = lib.stream_obj().record_event() if side else None       # dy is ready here
        if ctx.needs_input_grad[0]:
            dx = torch.empty(x.shape, dtype=ctx.x_dtype, device=x.device)
            flags = dy_f32 | ((_is_f32(dx) << 1) if cd == HIFIC_BF16 else 0)
            wc = _wcache(weight, 1, (N, C, H, W, K, R, S, stride, pt, pl, pb, pr, pad_mode), cd, flags, w_scale)
            call("hific_conv2d_bwd_data", ptr(dy), ptr(weight), ptr(w_scale), ptr(dx), N, C, H, W, K, R, S, stride,
                 pt, pl, pb, pr, pad_mode, cd, flags, wsp, wsb, *wc, stream())

        def param_grads():
            nonlocal dw, db
            wsp_, wsb_ = _ws(x)                      # the workspace of the stream these launches go to
            if want_w:
                dwt, acc, dw = _grad_target(ctx.w_slot, weight)
                flags = (_is_f32(x) if cd == HIFIC_BF16 else 0) | (dy_f32 << 1)
                call("hific_conv2d_bwd_weight", ptr(x), ptr(dy), ptr(dwt), N, C, H, W, K, R, S, stride, pt, pl, pb, pr,
                     pad_mode, acc, cd, flags, wsp_, wsb_, stream())
            if want_b:
                dbt, acc, db = _grad_target(ctx.b_slot, weight.new_empty(K))
                call("hific_channel_sum", ptr(dy), ptr(dbt), N, K, dy.shape[2] * dy.shape[3], acc, lib.dtype_code(dy),
                     wsp_, wsb_, stream())
        if side:
            with _SideLaunch(ev, x, dy, key=ctx.w_slot if ctx.w_slot is not None else ctx.b_slot):
                param_grads()
        else:
            param_grads()
        _written(ctx.w_slot if ctx.needs_input_grad[1] else None, ctx.b_slot if want_b else None)
        return dx, dw, db, None, None, None, None, None, None, None


def conv2d(x, weight, bias, stride=1, pads=(0, 0, 0, 0), pad_mode=lib.PAD_ZERO, act=None, out_f32=False, w_scale=None,
           exact=False, bias_grad=True, x3=None):
    """`exact=True` (bf16 compute mode only): split-bf16 forward on float32 activations, float32 output - the exact-index
    chain (see set_exact_index); the backward pass is the ordinary bf16 one.  `x3`: the (hi, lo, hi) image of `x` when the
    caller already has it (then `x` may be the nominal bf16 activation)."""
    pt, pl, pb, pr = pads
    if exact and x3 is None and _compute_dtype == torch.bfloat16 and x.dtype != torch.float32:
        x = cast_grad(x, torch.float32)
    return Conv2dFn.apply(_c16(x), weight, bias, (stride, pt, pl, pb, pr, pad_mode), act, out_f32, w_scale, exact,
                          bias_grad, x3)


class ConvTranspose2dFn(Function):
    """nn.ConvTranspose2d semantics: weight [Cin, Cout, R, S]; geom = (stride, pad, outpad)."""

    @staticmethod
    def forward(ctx, x, weight, bias, geom, act, out_f32, exact=False, bias_grad=True):
        require_gpu(x, weight, bias)
        cd = _cd()
        ctx.bias_grad = bool(bias_grad)
        stride, pad, outpad = geom
        N, Ci, H, W = x.shape
        Ciw, Co, R, S = weight.shape
        assert Ciw == Ci
        OH = (H - 1) * stride - 2 * pad + R + outpad
        OW = (W - 1) * stride - 2 * pad + S + outpad
        recon = exact in ("recon", "recon_pair")                # see Conv2dFn
        lay = SPLIT_PAIR if (exact in ("pair", "recon_pair") and Ci >= 16) else SPLIT_3C
        exact = bool(exact) and cd == HIFIC_BF16
        ydt = torch.float32 if (cd == HIFIC_F32 or out_f32 or exact) else torch.bfloat16
        if cd == HIFIC_F32 and x.dtype != torch.float32:
            raise lib.HificError("float32 compute mode needs float32 activations")
        y = torch.empty((N, Co, OH, OW), dtype=ydt, device=x.device)
        wsp, wsb = _ws(x)
        if exact:
            x3, w3 = _split3_act(x, lay), split_weights.get(weight, transposed=True, layout=lay)
            Cx = x3.shape[1]
            flags = _exact_flags(lay, Ci)
            wc = _wcache(w3, 0, (N, Cx, H, W, Co, R, S, stride, pad, outpad), cd, flags & 0xff, None, transposed=True)
            call("hific_conv_transpose2d_fwd", ptr(x3), ptr(w3), ptr(bias), ptr(y), N, Cx, H, W, Co, R, S, stride, pad,
                 outpad, _act_code(act), cd, flags, wsp, wsb, *wc, stream())
        else:
            flags = 0
            if cd == HIFIC_BF16:
                flags = _is_f32(x) | (_is_f32(y) << 1)
            wc = _wcache(weight, 0, (N, Ci, H, W, Co, R, S, stride, pad, outpad), cd, flags, None, transposed=True)
            call("hific_conv_transpose2d_fwd", ptr(x), ptr(weight), ptr(bias), ptr(y), N, Ci, H, W, Co, R, S, stride, pad,
                 outpad, _act_code(act), cd, flags, wsp, wsb, *wc, stream())
        ctx.geom, ctx.act, ctx.cd, ctx.has_bias = geom, act, cd, bias is not None
        ctx.w_slot, ctx.b_slot = _slot(weight), _slot(bias)
        ctx.bf16_bwd = bool(exact and recon and torch.is_grad_enabled() and x.dtype == torch.float32)        # see Conv2dFn
        ctx.x_dtype = x.dtype
        xs = cast(x, torch.bfloat16) if ctx.bf16_bwd else x
        ctx.save_for_backward(xs, weight, y if act not in (None, "none") else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        stride, pad, outpad = ctx.geom
        cd = ctx.cd
        N, Ci, H, W = x.shape
        _, Co, R, S = weight.shape
        dy = _c16(dy)
        wsp, wsb = _ws(x)
        if y is not None:
            dz = torch.empty_like(dy)
            slope = 0.0 if ctx.act == "relu" else 0.2
            call("hific_act_bwd", ptr(dy), ptr(y), ptr(dz), dy.numel(), slope, lib.dtype_code(dy), stream())
            dy = dz
        if ctx.bf16_bwd and dy.dtype != torch.bfloat16:
            dy = cast(dy, torch.bfloat16)
        dy_f32 = _is_f32(dy) if cd == HIFIC_BF16 else 0
        dx = dw = db = None
        want_w, want_b = ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2] and ctx.bias_grad
        side = (want_w or want_b) and _use_side(ctx.w_slot if want_w else True, ctx.b_slot if want_b else True)
        ev = lib.stream_obj().record_event() if side else None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(x.shape, dtype=ctx.x_dtype, device=x.device)
            flags = dy_f32 | ((_is_f32(dx) << 1) if cd == HIFIC_BF16 else 0)
            wc = _wcache(weight, 1, (N, Ci, H, W, Co, R, S, stride, pad, outpad), cd, flags, None, transposed=True)
            call("hific_conv_transpose2d_bwd_data", ptr(dy), ptr(weight), ptr(dx), N, Ci, H, W, Co, R, S, stride, pad,
                 outpad, cd, flags, wsp, wsb, *wc, stream())

        def param_grads():
            nonlocal dw, db
            wsp_, wsb_ = _ws(x)
            if want_w:
                dwt, acc, dw = _grad_target(ctx.w_slot, weight)
                flags = (_is_f32(x) if cd == HIFIC_BF16 else 0) | (dy_f32 << 1)
                call("hific_conv_transpose2d_bwd_weight", ptr(x), ptr(dy), ptr(dwt), N, Ci, H, W, Co, R, S, stride, pad,
                     outpad, acc, cd, flags, wsp_, wsb_, stream())
            if want_b:
                dbt, acc, db = _grad_target(ctx.b_slot, weight.new_empty(Co))
                call("hific_channel_sum", ptr(dy), ptr(dbt), N, Co, dy.shape[2] * dy.shape[3], acc, lib.dtype_code(dy),
                     wsp_, wsb_, stream())
        if side:
            with _SideLaunch(ev, x, dy, key=ctx.w_slot if ctx.w_slot is not None else ctx.b_slot):
                param_grads()
        else:
            param_grads()
        _written(ctx.w_slot if ctx.needs_input_grad[1] else None, ctx.b_slot if want_b else None)
        return dx, dw, db, None, None, None, None, None


def conv_transpose2d(x, weight, bias, stride, pad, outpad, act=None, out_f32=False, exact=False, bias_grad=True):
    if exact and _compute_dtype == torch.bfloat16 and x.dtype != torch.float32:
        x = cast_grad(x, torch.float32)
    return ConvTranspose2dFn.apply(_c16(x), weight, bias, (stride, pad, outpad), act, out_f32, exact, bias_grad)


# ------------------------------------------------------------------------------------------------------
class ChannelNormFn(Function):
    """`prev_bias`: the bias of the convolution that produced `x` when this norm is its only consumer (see
    normalisation.channel.fuse_bias_grad): its gradient, sum_{n,hw} dx, then comes out of the norm's backward kernel and that
    convolution skips its own channel-sum pass."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, relu, prev_bias=None, resid=None):
        """`resid` (same shape / dtype as x): y = act(norm(x)) + resid in the same kernel (the residual add of a
        ResidualBlock, generator.py:44); its gradient is dy itself."""
        require_gpu(x, gamma, beta, resid)
        N, C, H, W = x.shape
        y = torch.empty_like(x)
        mean = torch.empty((N, H * W), dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        fused = False
        if resid is not None:
            assert resid.shape == x.shape and resid.dtype == x.dtype
            rc = lib.raw("hific_channelnorm_fwd_res")(ptr(x), ptr(gamma), ptr(beta), ptr(resid), ptr(y), ptr(mean), ptr(rstd), N,
                                                      C, H * W, float(eps), int(relu), lib.dtype_code(x), stream())
            if rc == 0:
                fused = True
            elif rc != -4:
                raise lib.HificError(f"hific_channelnorm_fwd_res failed (rc={rc})")
        if not fused:
            call("hific_channelnorm_fwd", ptr(x), ptr(gamma), ptr(beta), ptr(y), ptr(mean), ptr(rstd), N, C, H * W,
                 float(eps), int(relu), lib.dtype_code(x), stream())
            if resid is not None:
                y = _add(y, resid)
        ctx.has_resid = resid is not None
        ctx.relu = int(relu)
        ctx.g_slot, ctx.b_slot = _slot(gamma), _slot(beta)
        ctx.has_prev, ctx.p_slot = prev_bias is not None, _slot(prev_bias)
        ctx.save_for_backward(x, gamma, beta, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, mean, rstd = ctx.saved_tensors
        N, C, H, W = x.shape
        dy = dy.contiguous()
        if dy.dtype != x.dtype:
            raise lib.HificError("ChannelNorm backward: grad dtype mismatch")
        dx = torch.empty_like(x)
        dgt, acc_g, dg = _grad_target(ctx.g_slot, gamma)
        dbt, acc_b, db = _grad_target(ctx.b_slot, beta)
        assert acc_g == acc_b
        dpt, acc_p, dp = (None, 0, None)
        if ctx.has_prev and ctx.needs_input_grad[5]:
            dpt, acc_p, dp = _grad_target(ctx.p_slot, gamma.new_empty(C))
        wsp, wsb = _ws(x)
        call("hific_channelnorm_bwd", ptr(x), ptr(dy), ptr(gamma), ptr(beta), ptr(mean), ptr(rstd), ptr(dx), ptr(dgt),
             ptr(dbt), N, C, H * W, ctx.relu, acc_g, lib.dtype_code(x), wsp, wsb, ptr(dpt), acc_p, stream())
        _written(ctx.g_slot, ctx.b_slot, ctx.p_slot if dpt is not None else None)
        return dx, dg, db, None, None, dp, (dy if ctx.has_resid else None)


class ExactConvNormFn(Function):
    """One conv -> ChannelNorm[+ReLU][+residual] block of an exact chain with a plain-bf16 autograd graph:
        (y, x3_next) = ChannelNorm[+ReLU]( conv_exact(x3; W) + b ) [+ resid].
    `x` is the block's NOMINAL input (bf16 NCHW activation, or a float32 tensor) - saved for the weight gradient only;
    `x3` is its split-bf16 image (hi, lo, hi) / pair layout that the forward contraction really reads (hific_split3 layout,
    produced by the previous block's norm kernel).  The convolution writes float32 z to a temporary;
    hific_channelnorm_fwd_exact turns it into the nominal bf16 output y, the next block's x3 and bf16(z) for the norm's
    backward.  Backward = exactly what the plain bf16 mode runs (ChannelNorm backward with the conv's bias gradient fused,
    data gradient, weight gradient on the side stream): no float32 activation is stored, re-read or back-propagated.
    Used by the Encoder (exact-index chain, encoder.py:56-93) and, round 6, by the Generator under
    ops.set_exact_training / set_exact_reconstruction (generator.py:9-44,115-137):
      * geom of 3 entries (stride, pad, outpad): the convolution is an nn.ConvTranspose2d (weight [Cin, Cout, R, S]);
      * resid / resid3: the residual of a ResidualBlock's second norm - `resid` the nominal tensor (autograd edge: its
        gradient is dy), `resid3` its split image in layout `lay_res`, which the norm kernel adds as hi + lo."""

    @staticmethod
    def forward(ctx, x, x3, weight, bias, geom, gamma, beta, eps, relu, lay_in=SPLIT_3C, lay_out=SPLIT_3C, resid=None,
                resid3=None, lay_res=SPLIT_3C):
        require_gpu(x, x3, weight, bias, gamma, beta, resid, resid3)
        transposed = len(geom) == 3
        N, C, H, W = x.shape
        Cx = pair_channels(C) if lay_in == SPLIT_PAIR else 3 * C
        assert tuple(x3.shape) == (N, Cx, H, W) and x3.dtype == torch.bfloat16
        flags = _exact_flags(lay_in, C)
        wsp, wsb = _ws(x)
        if _split_in_pack(weight, lay_in, transposed):
            w3, flags = weight, flags | 32                   # the pack pass forms (hi, hi, lo) from the master weight
        else:
            w3 = split_weights.get(weight, transposed=transposed, layout=lay_in)
        if transposed:
            stride, pad, outpad = geom
            Cw, K, R, S = weight.shape
            assert Cw == C
            OH = (H - 1) * stride - 2 * pad + R + outpad
            OW = (W - 1) * stride - 2 * pad + S + outpad
            z = torch.empty((N, K, OH, OW), dtype=torch.float32, device=x.device)
            wc = _wcache(w3, 0, (N, Cx, H, W, K, R, S, stride, pad, outpad), HIFIC_BF16, flags & 0xff, None, transposed=True)
            call("hific_conv_transpose2d_fwd", ptr(x3), ptr(w3), ptr(bias), ptr(z), N, Cx, H, W, K, R, S, stride, pad,
                 outpad, lib.ACT_NONE, HIFIC_BF16, flags, wsp, wsb, *wc, stream())
        else:
            stride, pt, pl, pb, pr, pad_mode = geom
            K, Cw, R, S = weight.shape
            assert Cw == C
            OH = (H + pt + pb - R) // stride + 1
            OW = (W + pl + pr - S) // stride + 1
            z = torch.empty((N, K, OH, OW), dtype=torch.float32, device=x.device)
            wc = _wcache(w3, 0, (N, Cx, H, W, K, R, S, stride, pt, pl, pb, pr, pad_mode), HIFIC_BF16, flags & 0xff, None)
            call("hific_conv2d_fwd", ptr(x3), ptr(w3), None, ptr(bias), None, ptr(z), N, Cx, H, W, K, R, S, stride, pt, pl, pb,
                 pr, pad_mode, lib.ACT_NONE, HIFIC_BF16, flags, wsp, wsb, *wc, stream())
        zb = torch.empty((N, K, OH, OW), dtype=torch.bfloat16, device=x.device)
        y = torch.empty_like(zb)
        Kx = pair_channels(K) if lay_out == SPLIT_PAIR else 3 * K
        x3n = torch.empty((N, Kx, OH, OW), dtype=torch.bfloat16, device=x.device)
        mean = torch.empty((N, OH * OW), dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        if resid3 is not None:
            Kr = pair_channels(K) if lay_res == SPLIT_PAIR else 3 * K
            assert tuple(resid3.shape) == (N, Kr, OH, OW) and resid3.dtype == torch.bfloat16
            call("hific_channelnorm_fwd_exact_res", ptr(z), ptr(gamma), ptr(beta), ptr(resid3), lay_res, ptr(zb), ptr(y),
                 ptr(x3n), ptr(mean), ptr(rstd), N, K, OH * OW, float(eps), int(relu), lay_out, stream())
        else:
            call("hific_channelnorm_fwd_exact", ptr(z), ptr(gamma), ptr(beta), ptr(zb), ptr(y), ptr(x3n), ptr(mean), ptr(rstd),
                 N, K, OH * OW, float(eps), int(relu), lay_out, stream())
        ctx.geom, ctx.relu, ctx.has_bias, ctx.transposed = geom, int(relu), bias is not None, transposed
        ctx.has_resid = resid is not None
        ctx.w_slot, ctx.b_slot, ctx.g_slot, ctx.be_slot = _slot(weight), _slot(bias), _slot(gamma), _slot(beta)
        ctx.save_for_backward(x, weight, zb, gamma, beta, mean, rstd)
        ctx.mark_non_differentiable(x3n)
        # x3n never receives a gradient: without this the engine would MATERIALISE a zero gradient for it in every backward
        # (five fills of 23-377 MB per G-turn at batch 16 x 256^2: 0.1 ms and 0.73 GB of HBM writes for nothing)
        ctx.set_materialize_grads(False)
        return y, x3n

    @staticmethod
    def backward(ctx, dy, _unused):
        if dy is None:
            return (None,) * 14
        x, weight, zb, gamma, beta, mean, rstd = ctx.saved_tensors
        N, C, H, W = x.shape
        if ctx.transposed:
            stride, pad, outpad = ctx.geom
            _, K, R, S = weight.shape
        else:
            stride, pt, pl, pb, pr, pad_mode = ctx.geom
            K, _, R, S = weight.shape
        OH, OW = zb.shape[2], zb.shape[3]
        dy = _c16(dy)
        if dy.dtype != torch.bfloat16:
            dy = cast(dy, torch.bfloat16)
        cd = HIFIC_BF16
        # ---- ChannelNorm backward (bf16) with the convolution's bias gradient out of the same kernel ----------------
        dz = torch.empty_like(zb)
        dgt, acc_g, dg = _grad_target(ctx.g_slot, gamma)
        dbt, acc_b, dbe = _grad_target(ctx.be_slot, beta)
        assert acc_g == acc_b
        want_b = ctx.has_bias and ctx.needs_input_grad[3]
        dpt, acc_p, db = (None, 0, None)
        if want_b:
            dpt, acc_p, db = _grad_target(ctx.b_slot, gamma.new_empty(K))
        wsp, wsb = _ws(x)
        call("hific_channelnorm_bwd", ptr(zb), ptr(dy), ptr(gamma), ptr(beta), ptr(mean), ptr(rstd), ptr(dz), ptr(dgt),
             ptr(dbt), N, K, OH * OW, ctx.relu, acc_g, HIFIC_BF16, wsp, wsb, ptr(dpt), acc_p, stream())
        _written(ctx.g_slot, ctx.be_slot, ctx.b_slot if want_b else None)
        # ---- convolution backward: plain bf16 kernels on (x, dz) -------------------------------------------------
        dx = dw = None
        want_w = ctx.needs_input_grad[2]
        side = want_w and _use_side(ctx.w_slot)
        ev = lib.stream_obj().record_event() if side else None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            flags = (_is_f32(dx) << 1)
            if ctx.transposed:
                wc = _wcache(weight, 1, (N, C, H, W, K, R, S, stride, pad, outpad), cd, flags, None, transposed=True)
                call("hific_conv_transpose2d_bwd_data", ptr(dz), ptr(weight), ptr(dx), N, C, H, W, K, R, S, stride, pad,
                     outpad, cd, flags, wsp, wsb, *wc, stream())
            else:
                wc = _wcache(weight, 1, (N, C, H, W, K, R, S, stride, pt, pl, pb, pr, pad_mode), cd, flags, None)
                call("hific_conv2d_bwd_data", ptr(dz), ptr(weight), None, ptr(dx), N, C, H, W, K, R, S, stride, pt, pl, pb, pr,
                     pad_mode, cd, flags, wsp, wsb, *wc, stream())

        def wgrad():
            nonlocal dw
            wsp_, wsb_ = _ws(x)
            dwt, acc, dw = _grad_target(ctx.w_slot, weight)
            if ctx.transposed:
                call("hific_conv_transpose2d_bwd_weight", ptr(x), ptr(dz), ptr(dwt), N, C, H, W, K, R, S, stride, pad, outpad,
                     acc, cd, _is_f32(x), wsp_, wsb_, stream())
            else:
                call("hific_conv2d_bwd_weight", ptr(x), ptr(dz), ptr(dwt), N, C, H, W, K, R, S, stride, pt, pl, pb, pr,
                     pad_mode, acc, cd, _is_f32(x), wsp_, wsb_, stream())
        if want_w:
            if side:
                with _SideLaunch(ev, x, dz, key=ctx.w_slot):
                    wgrad()
            else:
                wgrad()
        _written(ctx.w_slot if want_w else None)
        return dx, None, dw, db, None, dg, dbe, None, None, None, None, (dy if ctx.has_resid else None), None, None


def exact_conv_norm(x, x3, weight, bias, stride, pads, pad_mode, gamma, beta, eps, relu, lay_in=SPLIT_3C, lay_out=SPLIT_3C,
                    resid=None, resid3=None, lay_res=SPLIT_3C):
    pt, pl, pb, pr = pads
    return ExactConvNormFn.apply(_c16(x), x3, weight, bias, (stride, pt, pl, pb, pr, pad_mode), gamma, beta, eps, relu,
                                 lay_in, lay_out, resid, resid3, lay_res)


def exact_conv_transpose_norm(x, x3, weight, bias, stride, pad, outpad, gamma, beta, eps, relu, lay_in=SPLIT_3C,
                              lay_out=SPLIT_3C):
    """The up-convolution blocks of the exact Generator chain (generator.py:115-137): nn.ConvTranspose2d -> ChannelNorm -> ReLU."""
    return ExactConvNormFn.apply(_c16(x), x3, weight, bias, (stride, pad, outpad), gamma, beta, eps, relu, lay_in,
                                 lay_out, None, None, SPLIT_3C)


class ConvNormFn(Function):
    """conv (or nn.ConvTranspose2d: geom of 3 entries) -> ChannelNorm[+ReLU][+residual] of the PLAIN path as one autograd
    node - the same kernels, in the same order, as Conv2dFn / ConvTranspose2dFn followed by ChannelNormFn with the bias
    gradient fused (bit-identical), but one Function.apply, one saved-tensor set and one backward call per block instead of
    two: the Generator's 23 conv -> norm pairs (src/network/generator.py:9-44,98-137) are 46 of the ~230 forward nodes of a
    training forward.  Host time only; round 6."""

    @staticmethod
    def forward(ctx, x, weight, bias, geom, gamma, beta, eps, relu, resid=None):
        require_gpu(x, weight, bias, gamma, beta, resid)
        cd = _cd()
        transposed = len(geom) == 3
        N, C, H, W = x.shape
        if cd == HIFIC_F32 and x.dtype != torch.float32:
            raise lib.HificError("float32 compute mode needs float32 activations")
        zdt = torch.float32 if cd == HIFIC_F32 else torch.bfloat16
        flags = _is_f32(x) if cd == HIFIC_BF16 else 0
        wsp, wsb = _ws(x)
        if transposed:
            stride, pad, outpad = geom
            Cw, K, R, S = weight.shape
            assert Cw == C
            OH = (H - 1) * stride - 2 * pad + R + outpad
            OW = (W - 1) * stride - 2 * pad + S + outpad
            z = torch.empty((N, K, OH, OW), dtype=zdt, device=x.device)
            wc = _wcache(weight, 0, (N, C, H, W, K, R, S, stride, pad, outpad), cd, flags, None, transposed=True)
            call("hific_conv_transpose2d_fwd", ptr(x), ptr(weight), ptr(bias), ptr(z), N, C, H, W, K, R, S, stride, pad,
                 outpad, lib.ACT_NONE, cd, flags, wsp, wsb, *wc, stream())
        else:
            stride, pt, pl, pb, pr, pad_mode = geom
            K, Cw, R, S = weight.shape
            assert Cw == C
            OH = (H + pt + pb - R) // stride + 1
            OW = (W + pl + pr - S) // stride + 1
            z = torch.empty((N, K, OH, OW), dtype=zdt, device=x.device)
            wc = _wcache(weight, 0, (N, C, H, W, K, R, S, stride, pt, pl, pb, pr, pad_mode), cd, flags, None)
            call("hific_conv2d_fwd", ptr(x), ptr(weight), None, ptr(bias), None, ptr(z), N, C, H, W, K, R, S, stride, pt, pl, pb,
                 pr, pad_mode, lib.ACT_NONE, cd, flags, wsp, wsb, *wc, stream())
        y = torch.empty_like(z)
        mean = torch.empty((N, OH * OW), dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        fused = False
        if resid is not None:
            assert resid.shape == z.shape and resid.dtype == z.dtype
            rc = lib.raw("hific_channelnorm_fwd_res")(ptr(z), ptr(gamma), ptr(beta), ptr(resid), ptr(y), ptr(mean), ptr(rstd), N,
                                                      K, OH * OW, float(eps), int(relu), lib.dtype_code(z), stream())
            if rc == 0:
                fused = True
            elif rc != -4:
                raise lib.HificError(f"hific_channelnorm_fwd_res failed (rc={rc})")
        if not fused:
            call("hific_channelnorm_fwd", ptr(z), ptr(gamma), ptr(beta), ptr(y), ptr(mean), ptr(rstd), N, K, OH * OW,
                 float(eps), int(relu), lib.dtype_code(z), stream())
            if resid is not None:
                y = _add(y, resid)
        ctx.geom, ctx.relu, ctx.has_bias, ctx.transposed, ctx.cd = geom, int(relu), bias is not None, transposed, cd
        ctx.has_resid = resid is not None
        ctx.w_slot, ctx.b_slot, ctx.g_slot, ctx.be_slot = _slot(weight), _slot(bias), _slot(gamma), _slot(beta)
        ctx.save_for_backward(x, weight, z, gamma, beta, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, z, gamma, beta, mean, rstd = ctx.saved_tensors
        N, C, H, W = x.shape
        cd = ctx.cd
        if ctx.transposed:
            stride, pad, outpad = ctx.geom
            _, K, R, S = weight.shape
        else:
            stride, pt, pl, pb, pr, pad_mode = ctx.geom
            K, _, R, S = weight.shape
        OH, OW = z.shape[2], z.shape[3]
        dy = _c16(dy)
        if dy.dtype != z.dtype:
            raise lib.HificError("ChannelNorm backward: grad dtype mismatch")
        # ---- ChannelNorm backward with the convolution's bias gradient out of the same kernel -------------------------
        dz = torch.empty_like(z)
        dgt, acc_g, dg = _grad_target(ctx.g_slot, gamma)
        dbt, acc_b, dbe = _grad_target(ctx.be_slot, beta)
        assert acc_g == acc_b
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        dpt, acc_p, db = (None, 0, None)
        if want_b:
            dpt, acc_p, db = _grad_target(ctx.b_slot, gamma.new_empty(K))
        wsp, wsb = _ws(x)
        call("hific_channelnorm_bwd", ptr(z), ptr(dy), ptr(gamma), ptr(beta), ptr(mean), ptr(rstd), ptr(dz), ptr(dgt),
             ptr(dbt), N, K, OH * OW, ctx.relu, acc_g, lib.dtype_code(z), wsp, wsb, ptr(dpt), acc_p, stream())
        _written(ctx.g_slot, ctx.be_slot, ctx.b_slot if want_b else None)
        # ---- convolution backward on (x, dz) ---------------------------------------------------------------------
        dx = dw = None
        want_w = ctx.needs_input_grad[1]
        side = want_w and _use_side(ctx.w_slot)
        ev = lib.stream_obj().record_event() if side else None
        dz_f32 = _is_f32(dz) if cd == HIFIC_BF16 else 0
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            flags = dz_f32 | ((_is_f32(dx) << 1) if cd == HIFIC_BF16 else 0)
            if ctx.transposed:
                wc = _wcache(weight, 1, (N, C, H, W, K, R, S, stride, pad, outpad), cd, flags, None, transposed=True)
                call("hific_conv_transpose2d_bwd_data", ptr(dz), ptr(weight), ptr(dx), N, C, H, W, K, R, S, stride, pad,
                     outpad, cd, flags, wsp, wsb, *wc, stream())
            else:
                wc = _wcache(weight, 1, (N, C, H, W, K, R, S, stride, pt, pl, pb, pr, pad_mode), cd, flags, None)
                call("hific_conv2d_bwd_data", ptr(dz), ptr(weight), None, ptr(dx), N, C, H, W, K, R, S, stride, pt, pl, pb, pr,
                     pad_mode, cd, flags, wsp, wsb, *wc, stream())

        def wgrad():
            nonlocal dw
            wsp_, wsb_ = _ws(x)
            dwt, acc, dw = _grad_target(ctx.w_slot, weight)
            flags = (_is_f32(x) if cd == HIFIC_BF16 else 0) | (dz_f32 << 1)
            if ctx.transposed:
                call("hific_conv_transpose2d_bwd_weight", ptr(x), ptr(dz), ptr(dwt), N, C, H, W, K, R, S, stride, pad, outpad,
                     acc, cd, flags, wsp_, wsb_, stream())
            else:
                call("hific_conv2d_bwd_weight", ptr(x), ptr(dz), ptr(dwt), N, C, H, W, K, R, S, stride, pt, pl, pb, pr,
                     pad_mode, acc, cd, flags, wsp_, wsb_, stream())
        if want_w:
            if side:
                with _SideLaunch(ev, x, dz, key=ctx.w_slot):
                    wgrad()
            else:
                wgrad()
        _written(ctx.w_slot if want_w else None)
        return dx, dw, db, None, dg, dbe, None, None, (dy if ctx.has_resid else None)


def conv_norm(x, conv, norm, resid=None):
    """One conv -> ChannelNorm block of the plain path (layers.HipConv2d / HipConvTranspose2d + channel.ChannelNorm2D whose
    bias gradient is fused: channel.fuse_bias_grad) as one node."""
    if conv.transposed:
        geom = (conv.stride[0], conv.padding[0], conv.output_padding[0])
    else:
        geom = (conv.stride[0],) + tuple(conv.pads) + (conv.hip_pad_mode,)
    return ConvNormFn.apply(_c16(x), conv.weight, conv.bias, geom, norm.gamma, norm.beta, norm.eps, norm.fuse_relu,
                            None if resid is None else resid.contiguous())


class AddSplitFn(Function):
    """(y, y3) = a + b for two activations of an exact chain: the sum is formed from the split images (hi + lo each), so it is
    float32-accurate; `a` / `b` are the nominal tensors (autograd edges only: both receive dy)."""

    @staticmethod
    def forward(ctx, a, a3, la, b, b3, lb, lo):
        require_gpu(a, a3, b, b3)
        N, C, H, W = a.shape
        assert a.shape == b.shape
        y = torch.empty((N, C, H, W), dtype=torch.bfloat16, device=a.device)
        Cy = pair_channels(C) if lo == SPLIT_PAIR else 3 * C
        y3 = torch.empty((N, Cy, H, W), dtype=torch.bfloat16, device=a.device)
        call("hific_add_split", ptr(a3), la, ptr(b3), lb, ptr(y), ptr(y3), lo, N, C, H * W, stream())
        ctx.dts = (a.dtype, b.dtype)
        ctx.mark_non_differentiable(y3)
        ctx.set_materialize_grads(False)
        return y, y3

    @staticmethod
    def backward(ctx, g, _unused):
        if g is None:
            return (None,) * 7
        ga = g if g.dtype == ctx.dts[0] else cast(g, ctx.dts[0])
        gb = g if g.dtype == ctx.dts[1] else cast(g, ctx.dts[1])
        return ga, None, None, gb, None, None, None


def add_split(a, a3, la, b, b3, lb, lo):
    return AddSplitFn.apply(a, a3, la, b, b3, lb, lo)


def split3_act(x, layout=SPLIT_3C):
    """float32 [N,C,H,W] -> bf16 split image (see _split3_act): the operand of the first exact convolution."""
    if x.dtype != torch.float32:
        x = cast(x.contiguous(), torch.float32)
    return _split3_act(x.contiguous(), layout)


def channel_norm(x, gamma, beta, eps=1e-3, relu=False, prev_bias=None, resid=None):
    return ChannelNormFn.apply(x.contiguous(), gamma, beta, eps, relu, prev_bias, None if resid is None else resid.contiguous())


# ------------------------------------------------------------------------------------------------------
def _add(a, b):
    o = torch.empty_like(a)
    call("hific_add", ptr(a), ptr(b), ptr(o), a.numel(), lib.dtype_code(a), stream())
    return o


class AddFn(Function):
    @staticmethod
    def forward(ctx, a, b):
        require_gpu(a, b)
        assert a.shape == b.shape and a.dtype == b.dtype
        return _add(a, b)

    @staticmethod
    def backward(ctx, g):
        return g, g


def add(a, b):
    return AddFn.apply(a.contiguous(), b.contiguous())


class TanhFn(Function):
    """tanh on the reconstruction (`normalize_input_image`, src/model.py:155-156)."""

    @staticmethod
    def forward(ctx, x):
        require_gpu(x)
        y = torch.empty_like(x)
        call("hific_tanh_fwd", ptr(x), ptr(y), x.numel(), lib.dtype_code(x), stream())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        g = g.contiguous()
        if g.dtype != y.dtype:
            g = cast(g, y.dtype)
        dx = torch.empty_like(y)
        call("hific_tanh_bwd", ptr(y), ptr(g), ptr(dx), y.numel(), lib.dtype_code(y), stream())
        return dx


class ScaleShiftFn(Function):
    """a * x + b (the [-1,1] -> [0,1] map of src/model.py:206-209)."""

    @staticmethod
    def forward(ctx, x, a, b):
        require_gpu(x)
        y = torch.empty_like(x)
        call("hific_scale_shift", ptr(x), ptr(y), x.numel(), float(a), float(b), lib.dtype_code(x), stream())
        ctx.a = float(a)
        return y

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        dx = torch.empty_like(g)
        call("hific_scale_shift", ptr(g), ptr(dx), g.numel(), ctx.a, 0.0, lib.dtype_code(g), stream())
        return dx, None, None


def tanh(x):
    return TanhFn.apply(x.contiguous())


def scale_shift(x, a, b):
    return ScaleShiftFn.apply(x.contiguous(), a, b)


class ForkFn(Function):
    """Explicit fan-out: returns two aliases of x; backward sums the two incoming gradients with the HIP add kernel
    (instead of leaving the accumulation to the autograd engine's ATen add)."""

    @staticmethod
    def forward(ctx, x):
        ctx.set_materialize_grads(False)
        return x.view_as(x), x.view_as(x)

    @staticmethod
    def backward(ctx, g1, g2):
        if g1 is None:
            return g2
        if g2 is None:
            return g1
        g1 = g1.contiguous()
        g2 = g2.contiguous()
        if g1.dtype != g2.dtype:
            if g1.dtype == torch.float32:
                g2 = cast(g2, torch.float32)
            else:
                g1 = cast(g1, torch.float32)
        return _add(g1, g2)


def fork(x):
    return ForkFn.apply(x)


def cast(x, dt):
    """dtype conversion on the HIP side (no autograd)."""
    require_gpu(x)
    o = torch.empty(x.shape, dtype=dt, device=x.device)
    call("hific_cast", ptr(x), lib.dtype_code(x), ptr(o), lib.dtype_code(o), x.numel(), stream())
    return o


class CastFn(Function):
    @staticmethod
    def forward(ctx, x, dt):
        ctx.src = x.dtype
        return cast(x.contiguous(), dt)

    @staticmethod
    def backward(ctx, g):
        return cast(g.contiguous(), ctx.src), None


def cast_grad(x, dt):
    if x.dtype == dt:
        return x
    return CastFn.apply(x, dt)


# ------------------------------------------------------------------------------------------------------
# entropy-model ops (all float32)
class AddNoiseFn(Function):
    """x + noise (noise drawn by the caller from torch's RNG, like the reference's uniform_)."""

    @staticmethod
    def forward(ctx, x, noise):
        require_gpu(x, noise)
        # the caller's noise may have been allocated on another stream than the one this runs on (branch streams) and is
        # usually dropped right after this call: keep the allocator from recycling it under the kernel
        noise.record_stream(torch.cuda.current_stream(noise.device))
        return _add(x, noise)

    @staticmethod
    def backward(ctx, g):
        return g, None


class RoundFn(Function):
    """floor(x - mean + 0.5) + mean (hard quantisation, reference _quantize mode='quantize'): zero gradient to x,
    identity gradient to mean."""

    @staticmethod
    def forward(ctx, x, mean):
        require_gpu(x, mean)
        o = torch.empty_like(x)
        call("hific_round_f32", ptr(x), ptr(mean), ptr(o), x.numel(), stream())
        ctx.has_mean = mean is not None
        return o

    @staticmethod
    def backward(ctx, g):
        return None, (g if ctx.has_mean else None)


class RoundSTFn(Function):
    """quantize_latents_st: forward = floor(x - mean + .5) + mean, backward: d/dx = 1, d/dmean = 0."""

    @staticmethod
    def forward(ctx, x, mean):
        require_gpu(x, mean)
        o = torch.empty_like(x)
        call("hific_round_f32", ptr(x), ptr(mean), ptr(o), x.numel(), stream())
        return o

    @staticmethod
    def backward(ctx, g):
        return g, None


class LowerBoundFn(Function):
    @staticmethod
    def forward(ctx, x, bound):
        require_gpu(x)
        o = torch.empty_like(x)
        call("hific_lower_bound_fwd", ptr(x), float(bound), ptr(o), x.numel(), stream())
        ctx.bound = float(bound)
        ctx.save_for_backward(x)
        return o

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        g = g.contiguous()
        dx = torch.empty_like(x)
        call("hific_lower_bound_bwd", ptr(x), ptr(g), ctx.bound, ptr(dx), x.numel(), stream())
        return dx, None


class GaussLikFn(Function):
    """latent_likelihood: max(Phi((.5-|x-m|)/s) - Phi(-(.5+|x-m|)/s), min_lik) with the LowerBoundToward gradient."""

    @staticmethod
    def forward(ctx, x, mean, scale, min_lik, logistic):
        require_gpu(x, mean, scale)
        lik = torch.empty_like(x)
        call("hific_gauss_lik_fwd", ptr(x), ptr(mean), ptr(scale), ptr(lik), x.numel(), float(min_lik), int(logistic),
             stream())
        ctx.min_lik, ctx.logistic = float(min_lik), int(logistic)
        ctx.save_for_backward(x, mean, scale)
        return lik

    @staticmethod
    def backward(ctx, g):
        x, mean, scale = ctx.saved_tensors
        g = g.contiguous()
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dm = torch.empty_like(x) if ctx.needs_input_grad[1] else None
        ds = torch.empty_like(x) if ctx.needs_input_grad[2] else None
        call("hific_gauss_lik_bwd", ptr(x), ptr(mean), ptr(scale), ptr(g), ptr(dx), ptr(dm), ptr(ds), x.numel(),
             ctx.min_lik, ctx.logistic, 0, 0, stream())
        return dx, dm, ds, None, None


def _ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr


class FactorizedLikFn(Function):
    """HyperpriorDensity.likelihood for x [N,C,H,W]; params = H_0..H_3, a_0..a_3, b_0..b_3."""

    @staticmethod
    def forward(ctx, x, min_lik, *params):
        require_gpu(x, *params)
        N, C, H, W = x.shape
        lik = torch.empty_like(x)
        call("hific_factorized_lik_fwd", ptr(x), _ptr_array(params), ptr(lik), N, C, H * W, float(min_lik), stream())
        ctx.min_lik = float(min_lik)
        ctx.p_slots = [_slot(p) for p in params]
        ctx.save_for_backward(x, *params)
        return lik

    @staticmethod
    def backward(ctx, g):
        x, *params = ctx.saved_tensors
        N, C, H, W = x.shape
        g = g.contiguous()
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        targets = [_grad_target(sl, p) for sl, p in zip(ctx.p_slots, params)]
        accs = {t[1] for t in targets}
        assert len(accs) == 1, "factorised-prior parameters must share one gradient state"
        wsp, wsb = _ws(x)
        call("hific_factorized_lik_bwd", ptr(x), _ptr_array(params), ptr(g), ptr(dx),
             _ptr_array([t[0] for t in targets]), N, C, H * W, ctx.min_lik, accs.pop(), wsp, wsb, stream())
        _written(*ctx.p_slots)
        return (dx, None, *[t[2] for t in targets])


class LogSumFn(Function):
    """mul * sum(log(p + eps)) -> 0-d tensor (the entropy estimate of src/hyperprior.py:80-93)."""

    @staticmethod
    def forward(ctx, p, eps, mul):
        require_gpu(p)
        out = torch.empty((), dtype=torch.float32, device=p.device)
        wsp, wsb = _ws(p)
        call("hific_logsum_fwd", ptr(p), ptr(out), p.numel(), float(eps), float(mul), wsp, wsb, stream())
        ctx.eps, ctx.mul = float(eps), float(mul)
        ctx.save_for_backward(p)
        return out

    @staticmethod
    def backward(ctx, g):
        (p,) = ctx.saved_tensors
        g = g.contiguous().float()
        dp = torch.empty_like(p)
        call("hific_logsum_bwd", ptr(p), ptr(g), ptr(dp), p.numel(), ctx.eps, ctx.mul, 0, stream())
        return dp, None, None


# ------------------------------------------------------------------------------------------------------
class LossCombineFn(Function):
    """total = ((penalty * nbpp + k_M * mse) + k_P * mean(lp)) [+ beta * g_loss] with penalty = q > target ? lambda_A : lambda_B
    evaluated on the device (hific_loss_combine_fwd): the loss composition of src/model.py:211-220,373-376 and the rate
    schedule of src/loss/losses.py:8-28 as ONE autograd node instead of ~12 zero-dimensional torch ops.  Returns (total, aux)
    with aux = [perceptual, penalty, weighted_rate, weighted_distortion] (no gradient) for logging."""

    @staticmethod
    def forward(ctx, mse, lp, nbpp, q, g_loss, kM, kP, lamA, lamB, target, beta):
        require_gpu(mse, lp, nbpp, q, g_loss)
        B = lp.numel()
        total = torch.empty((), dtype=torch.float32, device=mse.device)
        aux = torch.empty(4, dtype=torch.float32, device=mse.device)
        call("hific_loss_combine_fwd", ptr(mse), ptr(lp), B, ptr(nbpp), ptr(q), ptr(g_loss), float(kM), float(kP), float(lamA),
             float(lamB), float(target), float(beta), ptr(total), ptr(aux), stream())
        ctx.c = (B, float(kM), float(kP), float(beta), tuple(lp.shape), g_loss is not None)
        ctx.save_for_backward(aux)
        ctx.mark_non_differentiable(aux)
        return total, aux

    @staticmethod
    def backward(ctx, g, _aux):
        (aux,) = ctx.saved_tensors
        B, kM, kP, beta, lp_shape, has_g = ctx.c
        g = g.contiguous().float()
        grads = torch.empty(3 + B, dtype=torch.float32, device=aux.device)
        call("hific_loss_combine_bwd", ptr(g), ptr(aux), B, kM, kP, beta, ptr(grads), stream())
        return (grads[0], grads[3:].view(lp_shape), grads[1], None, grads[2] if has_g else None,
                None, None, None, None, None, None)


class MSEFn(Function):
    """mean((scale*a - scale*b)^2); a = reconstruction (compute dtype), b = float32 input image (no grad)."""

    @staticmethod
    def forward(ctx, a, b, scale):
        require_gpu(a, b)
        assert b.dtype == torch.float32 and a.shape == b.shape
        out = torch.empty((), dtype=torch.float32, device=a.device)
        wsp, wsb = _ws(a)
        call("hific_mse_fwd", ptr(a), ptr(b), ptr(out), a.numel(), float(scale), lib.dtype_code(a), wsp, wsb, stream())
        ctx.scale = float(scale)
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = g.contiguous().float()
        da = torch.empty_like(a)
        call("hific_mse_bwd", ptr(a), ptr(b), ptr(g), ptr(da), a.numel(), ctx.scale, lib.dtype_code(a), stream())
        return da, None, None


class BCELogitsFn(Function):
    """mean BCE-with-logits of z (float32) against a constant target (ones / zeros)."""

    @staticmethod
    def forward(ctx, z, target):
        require_gpu(z)
        out = torch.empty((), dtype=torch.float32, device=z.device)
        wsp, wsb = _ws(z)
        call("hific_bce_fwd", ptr(z), float(target), ptr(out), z.numel(), wsp, wsb, stream())
        ctx.target = float(target)
        ctx.save_for_backward(z)
        return out

    @staticmethod
    def backward(ctx, g):
        (z,) = ctx.saved_tensors
        g = g.contiguous().float()
        dz = torch.empty_like(z)
        call("hific_bce_bwd", ptr(z), ctx.target, ptr(g), ptr(dz), z.numel(), 0, stream())
        return dz, None


class LsqSigmoidFn(Function):
    """mean((sigmoid(z) - target)^2): the least-squares GAN term on the Discriminator's sigmoid output."""

    @staticmethod
    def forward(ctx, z, target):
        require_gpu(z)
        out = torch.empty((), dtype=torch.float32, device=z.device)
        wsp, wsb = _ws(z)
        call("hific_lsq_sigmoid_fwd", ptr(z), float(target), ptr(out), z.numel(), wsp, wsb, stream())
        ctx.target = float(target)
        ctx.save_for_backward(z)
        return out

    @staticmethod
    def backward(ctx, g):
        (z,) = ctx.saved_tensors
        g = g.contiguous().float()
        dz = torch.empty_like(z)
        call("hific_lsq_sigmoid_bwd", ptr(z), ctx.target, ptr(g), ptr(dz), z.numel(), 0, stream())
        return dz, None


def sigmoid(z):
    require_gpu(z)
    o = torch.empty_like(z)
    call("hific_sigmoid_f32", ptr(z), ptr(o), z.numel(), stream())
    return o


class UpsampleConcatFn(Function):
    """cat([img, nearest_upsample(ctx, f)], dim=1) without materialising the upsampled context separately."""

    @staticmethod
    def forward(ctx_, img, ctxt, f):
        require_gpu(img, ctxt)
        N, Ci, H, W = img.shape
        Cc = ctxt.shape[1]
        out = torch.empty((N, Ci + Cc, H, W), dtype=img.dtype, device=img.device)
        call("hific_upcat_fwd", ptr(img), ptr(ctxt), ptr(out), N, Ci, Cc, H, W, int(f), lib.dtype_code(img), stream())
        ctx_.dims = (N, Ci, Cc, H, W, int(f))
        return out

    @staticmethod
    def backward(ctx_, g):
        N, Ci, Cc, H, W, f = ctx_.dims
        g = g.contiguous()
        dimg = dctx = None
        if ctx_.needs_input_grad[0]:
            dimg = torch.empty((N, Ci, H, W), dtype=g.dtype, device=g.device)
        if ctx_.needs_input_grad[1]:
            dctx = torch.empty((N, Cc, H // f, W // f), dtype=g.dtype, device=g.device)
        call("hific_upcat_bwd", ptr(g), ptr(dimg), 0, N if dimg is not None else 0, ptr(dctx), N, Ci, Cc, H, W, f,
             lib.dtype_code(g), stream())
        return dimg, dctx, None


class UpsamplePairConcatFn(Function):
    """The Discriminator input of one turn from (real, gen, per-latent context maps): rows [0, B) = real images, [B, 2B) =
    generated images, image n reads the context map of latent n >> 1 - what the reference builds with torch.cat and
    repeat_interleave (src/model.py:176-179) and an upsample + cat (discriminator.py:36,75-77), in one gather kernel."""

    @staticmethod
    def forward(ctx_, real, gen, ctxt, f):
        require_gpu(real, gen, ctxt)
        B, Ci, H, W = real.shape
        assert gen.shape == real.shape and gen.dtype == real.dtype == ctxt.dtype and ctxt.shape[0] == B
        Cc = ctxt.shape[1]
        out = torch.empty((2 * B, Ci + Cc, H, W), dtype=real.dtype, device=real.device)
        call("hific_upcat_pair_fwd", ptr(real), ptr(gen), ptr(ctxt), ptr(out), B, Ci, Cc, H, W, int(f), lib.dtype_code(real),
             stream())
        ctx_.dims = (B, Ci, Cc, H, W, int(f))
        return out

    @staticmethod
    def backward(ctx_, g):
        B, Ci, Cc, H, W, f = ctx_.dims
        g = g.contiguous()
        dgen = dctx = None
        if ctx_.needs_input_grad[1]:
            dgen = torch.empty((B, Ci, H, W), dtype=g.dtype, device=g.device)
        if ctx_.needs_input_grad[2]:
            dctx = torch.empty((B, Cc, H // f, W // f), dtype=g.dtype, device=g.device)
        if dgen is not None or dctx is not None:
            call("hific_upcat_pair_bwd", ptr(g), ptr(dgen), ptr(dctx), B, Ci, Cc, H, W, f, lib.dtype_code(g), stream())
        return None, dgen, dctx, None


def spectral_norm_power_iteration(weight_orig, u, v, do_iter, eps=1e-12):
    """In place on the (u, v) buffers, no autograd (torch.nn.utils.spectral_norm semantics: one iteration per
    training-mode forward).  Returns a 2-element tensor [sigma, 1/sigma]."""
    require_gpu(weight_orig, u, v)
    K = weight_orig.shape[0]
    M = weight_orig.numel() // K
    sig = torch.empty(2, dtype=torch.float32, device=weight_orig.device)
    wsp, wsb = _ws(weight_orig)
    call("hific_spectral_norm_fwd", ptr(weight_orig), ptr(u), ptr(v), ptr(sig), K, M, int(do_iter), float(eps),
         wsp, wsb, stream())
    return sig


def spectral_norm_power_iteration_batch(layers, do_iter, eps=1e-12):
    """`layers`: [(weight_orig, u, v), ...] (<= 8): one power iteration each (in place on u, v; no autograd), all layers per
    launch (hific_spectral_norm_fwd_batch).  Returns one tensor per layer (views of one buffer): [sigma, 1/sigma] followed by
    copies of the post-iteration u (K) and v (M) - the snapshot SNConv2dFn / D1StageFn save for their backward instead of
    cloning the buffers (16 device copies per training cycle); `sn_uv(sig, u, v)` splits it."""
    n = len(layers)
    ws0, us, vs = zip(*layers)
    require_gpu(*ws0, *us, *vs)
    kk = [w.shape[0] for w in ws0]
    mm = [w.numel() // w.shape[0] for w in ws0]
    offs, tot = [], 0
    for k, m in zip(kk, mm):
        offs.append(tot); tot += 2 + k + m
    sig = torch.empty(tot, dtype=torch.float32, device=ws0[0].device)
    Ks = (ctypes.c_int * n)(*kk)
    Ms = (ctypes.c_int * n)(*mm)
    base = sig.data_ptr()
    sp = (ctypes.c_void_p * n)(*[base + 4 * o for o in offs])
    wsp, wsb = _ws(ws0[0])
    call("hific_spectral_norm_fwd_batch", _ptr_array(ws0), _ptr_array(us), _ptr_array(vs), sp, Ks, Ms, n,
         (1 if do_iter else 0) | 2, float(eps), wsp, wsb, stream())
    return [sig[o:o + 2 + k + m] for o, k, m in zip(offs, kk, mm)]


def sn_uv(sig, u, v):
    """(sigma pair, u, v) for a spectral-norm layer's backward: the snapshot carried behind sigma when `sig` came from
    spectral_norm_power_iteration_batch, else clones of the live buffers (which the next forward iterates in place)."""
    K, M = u.numel(), v.numel()
    if sig.numel() == 2 + K + M:
        return sig[:2], sig[2:2 + K], sig[2 + K:]
    return sig, u.clone(), v.clone()


# spectral-norm convolutions: 1/sigma in the conv epilogue + cached packs (flags bit 4); HIFIC_SN_EPI_SCALE=0: scaled packs
_SN_EPI_SCALE = 16 if os.environ.get("HIFIC_SN_EPI_SCALE", "1") not in ("0", "") else 0


class SNConv2dFn(Function):
    """Conv2d with weight = weight_orig / sigma(u, v).  `sig` = [sigma, 1/sigma] from the power iteration; (u, v)
    are the post-iteration buffers (treated as constants, as in torch's spectral_norm)."""

    @staticmethod
    def forward(ctx, x, weight_orig, bias, u, v, sig, geom, act, out_f32):
        require_gpu(x, weight_orig, bias, u, v, sig)
        cd = _cd()
        stride, pt, pl, pb, pr, pad_mode = geom
        N, C, H, W = x.shape
        K, _, R, S = weight_orig.shape
        OH = (H + pt + pb - R) // stride + 1
        OW = (W + pl + pr - S) // stride + 1
        ydt = torch.float32 if (cd == HIFIC_F32 or out_f32) else torch.bfloat16
        y = torch.empty((N, K, OH, OW), dtype=ydt, device=x.device)
        flags = 0
        if cd == HIFIC_BF16:
            flags = _is_f32(x) | (_is_f32(y) << 1)
        wsp, wsb = _ws(x)
        inv_sigma = sig[1:]
        # flags bit 4: 1/sigma multiplies the accumulator in the conv epilogue, so the packed weights are those of
        # weight_orig alone and live in the pack cache like every other layer's (re-packed once per optimizer step in the
        # batched pass) - the sigma-scaled pack was a 12 us launch per spectral-norm conv, forward and data gradient
        flags |= _SN_EPI_SCALE
        wc = _wcache(weight_orig, 0, (N, C, H, W, K, R, S, stride, pt, pl, pb, pr, pad_mode), cd, flags, None) \
            if _SN_EPI_SCALE else (None, 0, 0)
        call("hific_conv2d_fwd", ptr(x), ptr(weight_orig), ptr(inv_sigma), ptr(bias), None, ptr(y),
             N, C, H, W, K, R, S, stride, pt, pl, pb, pr, pad_mode, _act_code(act), cd, flags, wsp, wsb, *wc,
             stream())
        ctx.geom, ctx.act, ctx.cd = geom, act, cd
        ctx.w_slot, ctx.b_slot = _slot(weight_orig), _slot(bias)
        sig2, us, vs = sn_uv(sig, u, v)
        ctx.save_for_backward(x, weight_orig, us, vs, sig2, y if act not in (None, "none") else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight_orig, u, v, sig, y = ctx.saved_tensors
        stride, pt, pl, pb, pr, pad_mode = ctx.geom
        cd = ctx.cd
        N, C, H, W = x.shape
        K, _, R, S = weight_orig.shape
        dy = _c16(dy)
        wsp, wsb = _ws(x)
        if y is not None:
            dz = torch.empty_like(dy)
            slope = 0.0 if ctx.act == "relu" else 0.2
            call("hific_act_bwd", ptr(dy), ptr(y), ptr(dz), dy.numel(), slope, lib.dtype_code(dy), stream())
            dy = dz
        dy_f32 = _is_f32(dy) if cd == HIFIC_BF16 else 0
        inv_sigma = sig[1:]
        dx = dw = db = None
        want_w, want_b = ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        side = (want_w or want_b) and _use_side(ctx.w_slot if want_w else True, ctx.b_slot if want_b else True)
        ev = lib.stream_obj().record_event() if side else None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            flags = dy_f32 | ((_is_f32(dx) << 1) if cd == HIFIC_BF16 else 0) | _SN_EPI_SCALE
            wc = _wcache(weight_orig, 1, (N, C, H, W, K, R, S, stride, pt, pl, pb, pr, pad_mode), cd, flags, None) \
                if _SN_EPI_SCALE else (None, 0, 0)
            call("hific_conv2d_bwd_data", ptr(dy), ptr(weight_orig), ptr(inv_sigma), ptr(dx), N, C, H, W, K, R, S,
                 stride, pt, pl, pb, pr, pad_mode, cd, flags, wsp, wsb, *wc, stream())

        def param_grads():
            nonlocal dw, db
            wsp_, wsb_ = _ws(x)
            if want_w:
                dws = torch.empty_like(weight_orig)       # gradient w.r.t. the normalised weight (stream-local scratch)
                flags = (_is_f32(x) if cd == HIFIC_BF16 else 0) | (dy_f32 << 1)
                call("hific_conv2d_bwd_weight", ptr(x), ptr(dy), ptr(dws), N, C, H, W, K, R, S, stride, pt, pl, pb, pr,
                     pad_mode, 0, cd, flags, wsp_, wsb_, stream())
                dwt, acc, dw = _grad_target(ctx.w_slot, weight_orig)
                M = weight_orig.numel() // K
                call("hific_spectral_norm_bwd", ptr(dws), ptr(weight_orig), ptr(u), ptr(v), ptr(sig), ptr(dwt), K, M,
                     acc, wsp_, wsb_, stream())
            if want_b:
                dbt, acc, db = _grad_target(ctx.b_slot, weight_orig.new_empty(K))
                call("hific_channel_sum", ptr(dy), ptr(dbt), N, K, dy.shape[2] * dy.shape[3], acc, lib.dtype_code(dy),
                     wsp_, wsb_, stream())
        if side:
            with _SideLaunch(ev, x, dy, u, v, sig, key=ctx.w_slot if ctx.w_slot is not None else ctx.b_slot):
                param_grads()
        else:
            param_grads()
        _written(ctx.w_slot if ctx.needs_input_grad[1] else None, ctx.b_slot if ctx.needs_input_grad[2] else None)
        return dx, dw, db, None, None, None, None, None, None


_D1_FUSED = os.environ.get("HIFIC_D1_FUSED", "1") not in ("0", "")


def d1_stage_on():
    """The Discriminator's input gather + first convolution as ONE autograd node (D1StageFn); HIFIC_D1_FUSED=0: the two nodes."""
    return _D1_FUSED


def set_d1_stage(on):
    global _D1_FUSED
    _D1_FUSED = bool(on)


class D1StageFn(Function):
    """UpsamplePairConcatFn + SNConv2dFn of the Discriminator's first layer (src/model.py:176-179, src/network/discriminator.py:
    36,53,75-78) as one node.  Forward: the same two kernels.  Backward: the 15-channel data gradient of the convolution on the
    258 x 258 padded plane is never formed - its consumers are (a) the generated images' 3 channels of the second half of the
    batch, computed by a data gradient restricted to those (G-turn only: the D-turn detaches them), and (b) the block sums over
    the 16 x 16 upsampling cells of the 12 context channels, which commute with the convolution's taps and come from window
    sums of the output gradient (hific_d1_ctx_grad).  Parameter gradients as in SNConv2dFn."""

    @staticmethod
    def forward(ctx, real, gen, ctxt, f, weight_orig, bias, u, v, sig, geom, act):
        require_gpu(real, gen, ctxt, weight_orig, bias, u, v, sig)
        B, Ci, H, W = real.shape
        assert gen.shape == real.shape and gen.dtype == real.dtype == ctxt.dtype and ctxt.shape[0] == B
        Cc = ctxt.shape[1]
        stride, pt, pl, pb, pr, pad_mode = geom
        K, C, R, S = weight_orig.shape
        assert C == Ci + Cc and (R, S, stride, pt, pl, pb, pr) == (4, 4, 2, 1, 1, 1, 1) and pad_mode == lib.PAD_REFLECT
        x = torch.empty((2 * B, C, H, W), dtype=real.dtype, device=real.device)
        call("hific_upcat_pair_fwd", ptr(real), ptr(gen), ptr(ctxt), ptr(x), B, Ci, Cc, H, W, int(f), lib.dtype_code(real),
             stream())
        cd = _cd()
        N = 2 * B
        OH, OW = (H + 2 - 4) // 2 + 1, (W + 2 - 4) // 2 + 1
        ydt = torch.float32 if cd == HIFIC_F32 else torch.bfloat16
        y = torch.empty((N, K, OH, OW), dtype=ydt, device=x.device)
        flags = (_is_f32(x) | (_is_f32(y) << 1)) if cd == HIFIC_BF16 else 0
        flags |= _SN_EPI_SCALE
        wsp, wsb = _ws(x)
        wc = _wcache(weight_orig, 0, (N, C, H, W, K, R, S, stride, pt, pl, pb, pr, pad_mode), cd, flags, None) \
            if _SN_EPI_SCALE else (None, 0, 0)
        call("hific_conv2d_fwd", ptr(x), ptr(weight_orig), ptr(sig[1:]), ptr(bias), None, ptr(y),
             N, C, H, W, K, R, S, stride, pt, pl, pb, pr, pad_mode, _act_code(act), cd, flags, wsp, wsb, *wc, stream())
        ctx.geom, ctx.act, ctx.cd, ctx.dims = geom, act, cd, (B, Ci, Cc, H, W, int(f))
        ctx.w_slot, ctx.b_slot = _slot(weight_orig), _slot(bias)
        sig2, us, vs = sn_uv(sig, u, v)
        ctx.save_for_backward(x, weight_orig, us, vs, sig2, y if act not in (None, "none") else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight_orig, u, v, sig, y = ctx.saved_tensors
        stride, pt, pl, pb, pr, pad_mode = ctx.geom
        cd = ctx.cd
        B, Ci, Cc, H, W, f = ctx.dims
        N, C = 2 * B, Ci + Cc
        K, _, R, S = weight_orig.shape
        dy = dy.contiguous()
        if y is not None:
            dz = torch.empty_like(dy)
            slope = 0.0 if ctx.act == "relu" else 0.2
            call("hific_act_bwd", ptr(dy), ptr(y), ptr(dz), dy.numel(), slope, lib.dtype_code(dy), stream())
            dy = dz
        dy_f32 = _is_f32(dy) if cd == HIFIC_BF16 else 0
        inv_sigma = sig[1:]
        dgen = dctx = dw = db = None
        want_w, want_b = ctx.needs_input_grad[4], ctx.needs_input_grad[5]
        side = (want_w or want_b) and _use_side(ctx.w_slot if want_w else True, ctx.b_slot if want_b else True)
        ev = lib.stream_obj().record_event() if side else None
        if ctx.needs_input_grad[1]:
            # generated images: data gradient of the second half of the batch for the image channels only
            wsp, wsb = _ws(x)
            w_gen = weight_orig[:, :Ci].contiguous()
            dgen = torch.empty((B, Ci, H, W), dtype=x.dtype, device=x.device)
            flags = dy_f32 | ((_is_f32(dgen) << 1) if cd == HIFIC_BF16 else 0) | _SN_EPI_SCALE
            call("hific_conv2d_bwd_data", ptr(dy[B:]), ptr(w_gen), ptr(inv_sigma), ptr(dgen), B, Ci, H, W, K, R, S,
                 stride, pt, pl, pb, pr, pad_mode, cd, flags, wsp, wsb, None, 0, 0, stream())
        if ctx.needs_input_grad[2]:
            dctx = torch.empty((B, Cc, H // f, W // f), dtype=dy.dtype, device=x.device)
            wsp, wsb = _ws(x)
            call("hific_d1_ctx_grad", ptr(dy), ptr(weight_orig), ptr(inv_sigma), ptr(dctx), B, K, Ci, Cc, H, W, f,
                 lib.dtype_code(dy), wsp, wsb, stream())
            if dctx.dtype != x.dtype:
                dctx = dctx.to(x.dtype)

        def param_grads():
            nonlocal dw, db
            wsp_, wsb_ = _ws(x)
            if want_w:
                dws = torch.empty_like(weight_orig)       # gradient w.r.t. the normalised weight (stream-local scratch)
                flags = (_is_f32(x) if cd == HIFIC_BF16 else 0) | (dy_f32 << 1)
                call("hific_conv2d_bwd_weight", ptr(x), ptr(dy), ptr(dws), N, C, H, W, K, R, S, stride, pt, pl, pb, pr,
                     pad_mode, 0, cd, flags, wsp_, wsb_, stream())
                dwt, acc, dw = _grad_target(ctx.w_slot, weight_orig)
                M = weight_orig.numel() // K
                call("hific_spectral_norm_bwd", ptr(dws), ptr(weight_orig), ptr(u), ptr(v), ptr(sig), ptr(dwt), K, M,
                     acc, wsp_, wsb_, stream())
            if want_b:
                dbt, acc, db = _grad_target(ctx.b_slot, weight_orig.new_empty(K))
                call("hific_channel_sum", ptr(dy), ptr(dbt), N, K, dy.shape[2] * dy.shape[3], acc, lib.dtype_code(dy),
                     wsp_, wsb_, stream())
        if side:
            with _SideLaunch(ev, x, dy, u, v, sig, key=ctx.w_slot if ctx.w_slot is not None else ctx.b_slot):
                param_grads()
        else:
            param_grads()
        _written(ctx.w_slot if want_w else None, ctx.b_slot if want_b else None)
        return None, dgen, dctx, None, dw, db, None, None, None, None, None


def adam_step(p, g, m, v, lr, beta1, beta2, eps, step, grad_scale=1.0):
    require_gpu(p, g, m, v)
    call("hific_adam_step", ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), float(lr), float(beta1), float(beta2),
         float(eps), int(step), float(grad_scale), stream())


def adam_prepare(step_dev, bc_dev, beta1, beta2):
    """++step (device int32), bias corrections -> bc_dev[0..1] (device float32): graph-replayable Adam (optim.FusedAdam)."""
    require_gpu(step_dev, bc_dev)
    call("hific_adam_prepare", ptr(step_dev), ptr(bc_dev), float(beta1), float(beta2), stream())


def adam_apply(p, g, m, v, lr, beta1, beta2, eps, bc_dev, grad_scale=1.0):
    require_gpu(p, g, m, v, bc_dev)
    call("hific_adam_apply", ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), float(lr), float(beta1), float(beta2),
         float(eps), ptr(bc_dev), float(grad_scale), stream())


# ---- EVALUATION path: device half of `compress` (int32 symbols + table indices for the host rANS coder) -------------
SCALES_MIN = 0.11        # src/compression/prior_model.py:19


def prior_symbols_and_indices(latents, means, scales, scale_table, scales_min=SCALES_MIN):
    """`PriorEntropyModel.compress` up to the host handoff (prior_model.py:178-181 with compute_indices :148-156):
    returns (symbols, indices), int32, same shape as `latents`."""
    x, m, sc = latents.contiguous().float(), means.contiguous().float(), scales.contiguous().float()
    tab = scale_table.to(x.device).contiguous().float()
    require_gpu(x, m, sc, tab)
    if m.shape != x.shape or sc.shape != x.shape:
        raise lib.HificError("prior_symbols_and_indices: means/scales must have the shape of the latents")
    sym = torch.empty(x.shape, dtype=torch.int32, device=x.device)
    idx = torch.empty(x.shape, dtype=torch.int32, device=x.device)
    call("hific_prior_symbols", ptr(x), ptr(m), ptr(sc), ptr(tab), tab.numel(), float(scales_min), ptr(sym), ptr(idx),
         x.numel(), stream())
    return sym, idx


def hyper_symbols_and_indices(hyperlatents):
    """`HyperpriorEntropyModel.compress` up to the host handoff (hyperprior_model.py:160-169): symbols = floor(z+.5),
    indices = channel number, both int32 (N,C,H,W)."""
    z = hyperlatents.contiguous().float()
    require_gpu(z)
    N, C, H, W = z.shape
    sym = torch.empty(z.shape, dtype=torch.int32, device=z.device)
    idx = torch.empty(z.shape, dtype=torch.int32, device=z.device)
    call("hific_hyper_symbols", ptr(z), ptr(sym), ptr(idx), N, C, H * W, stream())
    return sym, idx

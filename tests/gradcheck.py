"""How gradients are compared with the oracle in float32 parity mode (used by test_gpu_modules.py and
test_gpu_fullsize_backward.py).

Bar: max |g - g_ref| <= 1e-3 x max |g_ref| per tensor against the oracle run in float32 - the arithmetic the reference
executes.  Two effects can push a tensor over that bar without any arithmetic being wrong; both are judged explicitly:

1. Conditioning.  Where the oracle's own float32 run deviates from its float64 run by more than the bar (the likelihood
   gradient d/d sigma of Phi((.5-a)/sigma) - Phi(-(.5+a)/sigma) near the 1e-9 floor is a difference of nearly equal
   numbers: 7.9e-3 on Hyperprior.synthesis_std.conv1.weight at batch 4 x 256^2), the float64 oracle is the arbiter: the
   device must be as close to it as 3x the float32 oracle's own deviation.
2. ReLU sign ties - the backward-pass sibling of the quantiser's rounding ties.  A pre-activation within float32 summation
   noise of 0 gets the mask 1 on one side and 0 on the other; the gradient through that ONE element then differs entirely.
   In the hyper-prior nets (4x4 ... 16x16 planes, a weight gradient is a sum over only 256 ... 4096 positions) one such
   element moves the gradients of ONE output channel by ~1/positions: measured 1.2e-3 ... 4e-3 on synthesis_{mu,std}
   .conv{1,2} and analysis_net.conv1 at the benchmark shapes, while float32 and float64 oracles agree to 1e-6 there.  The
   signature is localisation: every element beyond the bar lies in a handful of output channels.  Accepted iff the elements
   beyond the bar are confined to <= 8 slices along the (in or out) channel dimension and stay below 3e-2 - AND, when the
   caller traced the oracle's activations (`relu_trace`, round 4), a pre-activation of that very channel within float32
   summation noise of zero is actually found (located and printed): localisation alone is a plausible story, the located
   element is the proof."""
import contextlib

import torch

# |z| <= TIE_REL x max|z| of the layer counts as "within summation noise of zero": float32 accumulation over ~3e3-3e4 products
# of O(1e-2..1) terms in another order moves a sum by up to ~1e-6 of the layer's scale (located at the benchmark shapes, round 4:
# 3.7e-8 ... 2.5e-7)
TIE_REL = 1e-6


class _Trace:
    """Per activation call of the oracle: (kind, channels, per-channel min |z| over batch and positions, max |z|)."""

    def __init__(self):
        self.calls = []

    def record(self, kind, z):
        if z.ndim == 4:
            a = z.detach().abs()
            self.calls.append((kind, z.shape[1], a.amin(dim=(0, 2, 3)).double(), float(a.max())))

    def find_tie(self, nchan, channels):
        """-> (call index, channel, |z|, layer scale) of the smallest pre-activation among `channels` over all traced
        activation calls with `nchan` channels, or None."""
        best = None
        for i, (kind, C, mn, scale) in enumerate(self.calls):
            if C != nchan:
                continue
            for c in channels:
                v = float(mn[c]) / max(scale, 1e-30)
                if best is None or v < best[2]:
                    best = (i, int(c), v, scale, kind)
        return best


class _FProxy:
    """torch.nn.functional with relu / leaky_relu recorded (the oracle module's `F` is swapped for this during a trace)."""

    def __init__(self, real, trace):
        self._real, self._trace = real, trace

    def __getattr__(self, name):
        return getattr(self._real, name)

    def relu(self, z, *a, **k):
        self._trace.record("relu", z)
        return self._real.relu(z, *a, **k)

    def leaky_relu(self, z, *a, **k):
        self._trace.record("leaky_relu", z)
        return self._real.leaky_relu(z, *a, **k)


@contextlib.contextmanager
def relu_trace(oracle_module):
    """`with relu_trace(O) as tr:` - every F.relu / F.leaky_relu input of the oracle's forward inside is summarised in tr."""
    tr = _Trace()
    real = oracle_module.F
    oracle_module.F = _FProxy(real, tr)
    try:
        yield tr
    finally:
        oracle_module.F = real


def relerr(a, b):
    return float((a.double() - b.double()).abs().max()) / max(float(b.abs().max()), 1e-30)


def _affected(got, ref, tol):
    """(number of channel slices, their indices, the size of that channel dimension): the slices - along dim 0 or dim 1,
    whichever has fewer; elements for 1-d tensors - that contain an element whose error exceeds tol x scale."""
    scale = max(float(ref.abs().max()), 1e-30)
    viol = (got.double() - ref.double()).abs() > tol * scale
    if viol.ndim <= 1 or viol.shape[0] == 1 and viol.ndim == 2:
        idx = viol.flatten().nonzero().flatten().tolist()
        return len(idx), idx, viol.numel()
    if viol.ndim == 4 and viol.shape[0] == 1 and viol.shape[2] == 1:       # (1, C, 1, 1) affine parameters
        idx = viol.flatten().nonzero().flatten().tolist()
        return len(idx), idx, viol.numel()
    i0 = viol.flatten(1).any(dim=1).nonzero().flatten().tolist()
    i1 = viol.transpose(0, 1).flatten(1).any(dim=1).nonzero().flatten().tolist()
    return (len(i0), i0, viol.shape[0]) if len(i0) <= len(i1) else (len(i1), i1, viol.shape[1])


def _affected_channels(got, ref, tol):
    return _affected(got, ref, tol)[0]


def check_grads(got, ref32, exact=None, tol=1e-3, what="", noise_factor=3.0, tie_channels=8, tie_cap=3e-2, trace=None):
    """got / ref32: {name: tensor}; exact: None, a {name: float64 tensor} dict, or a zero-argument callable returning one
    (only evaluated if some tensor misses `tol` against ref32); trace: a relu_trace() of the oracle forward that produced
    ref32 - then a "sign tie" is only accepted with the located near-zero pre-activation.  Returns (worst error vs ref32,
    {name: how it was judged})."""
    rows = sorted(((relerr(got[k], g), k) for k, g in ref32.items()), reverse=True)
    print(f"  [{what}] {len(rows)} tensors vs float32 oracle; worst: " + "; ".join(f"{k} {e:.2e}" for e, k in rows[:5]))
    miss = [(e, k) for e, k in rows if not e < tol]
    judged, bad = {}, []
    ex = None
    for e, k in miss:
        nch, chans, cdim = _affected(got[k], ref32[k], tol)
        if nch <= tie_channels and e <= tie_cap:
            proof = ""
            if trace is not None:
                hit = trace.find_tie(cdim, chans)
                if hit is None or not hit[2] <= TIE_REL:
                    bad.append((k, e, f"confined to channel slices {chans} but no pre-activation of those channels lies within "
                                      f"{TIE_REL:g} of zero (closest: {hit})"))
                    continue
                proof = (f"; located: {hit[4]} call #{hit[0]}, channel {hit[1]}, |z| = {hit[2]:.2e} x the layer's max |z| "
                         f"({hit[3]:.3g})")
            judged[k] = (f"sign ties: {e:.2e}, beyond-bar elements confined to channel slice(s) {chans} of "
                         f"{tuple(ref32[k].shape)}{proof}")
            print(f"    {k}: {judged[k]}")
            continue
        if exact is None:
            bad.append((k, e, f"{nch} channel slices affected"))
            continue
        if ex is None:
            ex = exact() if callable(exact) else exact
        e_dev, e_o32 = relerr(got[k], ex[k]), relerr(ref32[k], ex[k])
        bar = max(tol, noise_factor * e_o32)
        judged[k] = f"conditioning: device vs f64 {e_dev:.2e}, f32 oracle vs f64 {e_o32:.2e}, bar {bar:.2e}"
        print(f"    {k}: device vs f32 oracle {e:.2e} ({nch} channel slices); {judged[k]}")
        if not e_dev <= bar:
            bad.append((k, e_dev, bar))
    assert not bad, f"{what}: {bad}"
    return rows[0][0], judged

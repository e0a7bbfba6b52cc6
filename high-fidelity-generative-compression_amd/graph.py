"""hipGraph capture of a whole training cycle (SURVEY section 7 "hipGraph the step").

One G-turn + D-turn cycle is ~900 kernel launches issued from Python through autograd and ctypes: 18-26 ms of host time
per cycle against 25-30 ms of GPU time - every kernel-side gain below that floor is invisible.  `GraphedStep` runs the
step function a few times eagerly (so every persistent buffer exists: parameter arenas, packed-weight cache, split-weight
images, per-stream workspaces, pack job tables), captures ONE call into a hipGraph (torch.cuda.CUDAGraph = hipGraph on
ROCm, including the side / branch streams forked and joined inside the step and the whole autograd backward), and replays
it: one host call per cycle.

What makes the step capturable (each was a host-side decision baked into kernel arguments):
  * Adam's step count and bias corrections live in device memory (optim.FusedAdam -> hific_adam_prepare / _apply);
  * the rate-penalty branch is evaluated on the device (Model(device_rate_select=True)), no `.item()`;
  * overwrite-vs-accumulate of every gradient slot (GradSlot.fresh) is a fixed pattern once the cycle ends with the
    optimizers' zero_grad(): the capture records the steady-state pattern;
  * the batched weight re-pack (ops.WeightPackCache) is found stale at the same two places of every cycle and is part of
    the graph; its job tables are built during the warm-up;
  * random numbers: torch's device generators are graph-safe (philox offset advanced per replay); generators other than
    the default one must be passed in `generators=` to be registered with the graph.
What is frozen at capture: learning rate, the schedule values of lambda / target rate (src/helpers/utils.py:64-72 change at
step 50 000: re-capture then), log_interval bookkeeping (capture with writeout=False).  The captured function must not
synchronise with the host.  Results are bit-identical to eager execution (tests/test_gpu_zz_graph.py)."""
import torch


class GraphedStep:
    def __init__(self, fn, warmup=3, generators=(), stream=None):
        """fn: zero-argument callable performing one full cycle on persistent state; its return value (tensors living in
        the graph's memory pool) is returned by every replay."""
        self.fn = fn
        self.stream = stream or torch.cuda.Stream()
        cur = torch.cuda.current_stream()
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            for _ in range(max(1, warmup)):           # real steps: they advance the training state
                fn()
        cur.wait_stream(self.stream)
        torch.cuda.synchronize()
        # the device is idle: drop the cross-stream ordering events of the warm-up (a capturing stream may only wait on
        # events recorded inside the capture)
        from . import ops
        ops.pack_cache.settle()
        ops.wait_late_params()
        self.graph = torch.cuda.CUDAGraph()
        for g in generators:
            self.graph.register_generator_state(g)
        # "relaxed": hipFuncSetAttribute (dynamic LDS size, issued next to every launch by the conv planner) and the engine
        # thread's launches are not stream operations the capture needs to police
        with torch.cuda.graph(self.graph, stream=self.stream, capture_error_mode="relaxed"):
            self.out = fn()
        # the graph holds raw pointers into the packed-weight cache: its entries must outlive it (replays never touch the
        # cache's LRU bookkeeping)
        self._pins, self._pin_keys = ops.pack_cache.pin_all()
        self.replays = 0

    def close(self):
        """Releases the graph and its pins on the packed-weight cache (pinned entries are exempt from eviction: without this
        every re-capture would grow the cache past its cap)."""
        keys, self._pin_keys = getattr(self, "_pin_keys", None), None
        if keys:
            from . import ops
            ops.pack_cache.unpin(keys)
        self.graph = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __call__(self):
        from . import ops
        if ops.pack_cache.pins_broken != self._pins:
            raise RuntimeError("GraphedStep: a packed-weight buffer this graph reads was dropped (a parameter was moved, "
                               "re-created, or the cache was cleared) - capture the step again")
        if self.graph is None:
            raise RuntimeError("GraphedStep: closed")
        self.graph.replay()
        self.replays += 1
        return self.out

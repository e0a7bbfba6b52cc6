"""Drop-in for the reference's LPIPS perceptual loss as HiFIC uses it
(src/loss/perceptual_similarity/perceptual_loss.py:13-46 -> dist_model.py:105-113 -> networks_basic.py:61-88,
model='net-lin', net='alex' - or net='vgg', the reference's other documented backbone - version 0.1, spatial=False).

PerceptualLoss.forward(pred, target, normalize) -> (N,1,1,1), differentiable w.r.t. `pred` only (the backbone is
frozen and the target image carries no gradient on the HiFIC path, src/model.py:196-199).  The whole
forward/backward is ONE autograd node: both images run through AlexNet as one 2N batch on the gconv kernels, the
five taps are fused normalise/diff/lin/mean kernels (csrc/lpips.hip), the backward walks the gen half only.

Backbone weights: the reference downloads torchvision's ImageNet AlexNet at run time (pretrained_networks.py:59);
there is no network here, so the backbone is initialised from a fixed seed unless `load_backbone_state_dict` is
given real weights (keys `features.{0,3,6,8,10}.{weight,bias}`, torchvision layout).  The learned linear heads are
the reference's own v0.1 weights (loss/weights/lpips_alex_lin_v0.1.npz, made by tools/extract_lpips_lin.py).
"""
import os

import numpy as np
import torch
import torch.nn as nn
from torch.autograd import Function

from .. import lib, ops
from ..lib import call, ptr, stream

_HERE = os.path.dirname(os.path.abspath(__file__))

# Backbones (pretrained_networks.py:59-94 alexnet, :96-134 vgg16): per conv layer
#   (C_in, C_out, kernel, stride, pad, max-pool after the ReLU ('3s2' | '2s2' | None), tapped by LPIPS?, index in
#    torchvision's `features` Sequential)
NETS = {
    "alex": [(3, 64, 11, 4, 2, "3s2", True, 0), (64, 192, 5, 1, 2, "3s2", True, 3), (192, 384, 3, 1, 1, None, True, 6),
             (384, 256, 3, 1, 1, None, True, 8), (256, 256, 3, 1, 1, None, True, 10)],
    "vgg": [(3, 64, 3, 1, 1, None, False, 0), (64, 64, 3, 1, 1, "2s2", True, 2),
            (64, 128, 3, 1, 1, None, False, 5), (128, 128, 3, 1, 1, "2s2", True, 7),
            (128, 256, 3, 1, 1, None, False, 10), (256, 256, 3, 1, 1, None, False, 12), (256, 256, 3, 1, 1, "2s2", True, 14),
            (256, 512, 3, 1, 1, None, False, 17), (512, 512, 3, 1, 1, None, False, 19), (512, 512, 3, 1, 1, "2s2", True, 21),
            (512, 512, 3, 1, 1, None, False, 24), (512, 512, 3, 1, 1, None, False, 26), (512, 512, 3, 1, 1, None, True, 28)],
}
ALEX_CFG = [c[:5] + (c[5] is not None,) for c in NETS["alex"]]          # legacy names (tests, tools)
ALEX_FEATURE_IDX = [c[7] for c in NETS["alex"]]


def _conv_fwd(x, w, b, stride, pad, cd, out=None):
    N, C, H, W = x.shape
    K, _, R, S = w.shape
    OH = (H + 2 * pad - R) // stride + 1
    OW = (W + 2 * pad - S) // stride + 1
    y = out if out is not None else torch.empty((N, K, OH, OW), dtype=x.dtype, device=x.device)
    ws = lib.workspace(x.device)
    geom = (N, C, H, W, K, R, S, stride, pad, pad, pad, pad, lib.PAD_ZERO)
    wc = ops._wcache(w, 0, geom, cd, 0)                      # frozen backbone: packed once
    call("hific_conv2d_fwd", ptr(x), ptr(w), None, ptr(b), None, ptr(y), *geom, lib.ACT_RELU, cd, 0, ws.data_ptr(),
         ws.numel(), *wc, stream())
    return y


def _conv_bwd_data(dy, w, xshape, stride, pad, cd):
    N, C, H, W = xshape
    K, _, R, S = w.shape
    dx = torch.empty(xshape, dtype=dy.dtype, device=dy.device)
    ws = lib.workspace(dy.device)
    geom = (N, C, H, W, K, R, S, stride, pad, pad, pad, pad, lib.PAD_ZERO)
    wc = ops._wcache(w, 1, geom, cd, 0)
    call("hific_conv2d_bwd_data", ptr(dy), ptr(w), None, ptr(dx), *geom, cd, 0, ws.data_ptr(), ws.numel(), *wc, stream())
    return dx


def _maxpool(x, out=None):
    N, C, H, W = x.shape
    OH, OW = (H - 3) // 2 + 1, (W - 3) // 2 + 1
    y = out if out is not None else torch.empty((N, C, OH, OW), dtype=x.dtype, device=x.device)
    call("hific_maxpool3s2_fwd", ptr(x), ptr(y), N * C, H, W, lib.dtype_code(x), stream())
    return y


def _pool(f, kind):
    if kind == "3s2":
        return _maxpool(f)
    N, C, H, W = f.shape
    y = torch.empty((N, C, H // 2, W // 2), dtype=f.dtype, device=f.device)
    call("hific_maxpool2s2_fwd", ptr(f), ptr(y), N * C, H, W, lib.dtype_code(f), stream())
    return y


def _pooled_hw(kind, H, W):
    return ((H - 3) // 2 + 1, (W - 3) // 2 + 1) if kind == "3s2" else (H // 2, W // 2)


def _net_half(h, wb, cd, feats, lo, B, cfg):
    """Backbone features on the B images `h`; the feature map of conv layer li goes to rows [lo, lo+B) of the 2B-image
    buffer feats[li] (allocated here on first use).  Every output pixel is computed by the same instruction sequence
    whatever the batch size, so filling the two halves separately gives the bits of one 2B pass."""
    for li, (ci, co, k, s, p, pool, tap, idx) in enumerate(cfg):
        N, C, H, W = h.shape
        OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        if feats[li] is None:
            feats[li] = torch.empty((2 * B, co, OH, OW), dtype=h.dtype, device=h.device)
        f = feats[li][lo:lo + B]
        _conv_fwd(h, wb[2 * li], wb[2 * li + 1], s, p, cd, out=f)
        h = _pool(f, pool) if (pool and li < len(cfg) - 1) else f


class TargetFeatures:
    """Target-image half of the LPIPS feature maps, computed ahead of the loss (PerceptualLoss.prefetch_target): it
    depends on the input image only, so it can run on a second stream while the Encoder / Generator produce the other
    image."""

    def __init__(self, key, feats, event, target):
        # `target` keeps the tensor alive, so its address cannot be recycled for another batch while this is pending, and
        # the consumer matches by identity of the storage (data_ptr + version + shape of a live tensor)
        self.key, self.feats, self.event, self.target = key, feats, event, target
        self.stream_id = torch.cuda.current_stream(target.device).cuda_stream       # where `event` was recorded


def _target_key(target, normalize, cd, net="alex"):
    return (target.data_ptr(), target._version, tuple(target.shape), target.dtype, int(normalize), cd, net)


class LpipsFn(Function):
    @staticmethod
    def forward(ctx, pred, target, normalize, lins, pre, net, *wb):
        lib.require_gpu(pred, target, *lins, *wb)
        cfg = NETS[net]
        cdt = ops.get_compute_dtype()
        cd = lib.HIFIC_F32 if cdt == torch.float32 else lib.HIFIC_BF16
        B, _, H, W = pred.shape
        x = torch.empty((2 * B, 3, H, W), dtype=cdt, device=pred.device)
        # in0 = target, in1 = pred (perceptual_loss.py:40: self.model.forward(target, pred))
        call("hific_lpips_prep", ptr(target), 1 if target.dtype == torch.float32 else 0, ptr(pred),
             1 if pred.dtype == torch.float32 else 0, ptr(x), B, H * W, int(normalize), cd, stream())
        val = torch.empty(B, dtype=torch.float32, device=pred.device)
        ws = lib.workspace(pred.device)
        if pre is not None and pre.key == _target_key(target, normalize, cd, net):
            cur = torch.cuda.current_stream(pred.device)
            # same stream (the loss branch runs on the stream the prefetch ran on): stream order already holds - and a
            # stream waiting on its own event inside a hipGraph capture crashes hipStreamEndCapture (ROCm 7.2)
            if cur.cuda_stream != pre.stream_id:
                cur.wait_event(pre.event)
            feats = pre.feats
            for f in feats:
                f.record_stream(cur)
            _net_half(x[B:], wb, cd, feats, B, B, cfg)           # generated-image half only
        else:
            feats = [None] * len(cfg)
            h = x
            for li, (ci, co, k, s, p, pool, tap, idx) in enumerate(cfg):
                feats[li] = _conv_fwd(h, wb[2 * li], wb[2 * li + 1], s, p, cd)
                h = _pool(feats[li], pool) if (pool and li < len(cfg) - 1) else feats[li]
        ti = 0
        for li, (ci, co, k, s, p, pool, tap, idx) in enumerate(cfg):
            if not tap:
                continue
            f = feats[li]
            call("hific_lpips_tap_fwd", ptr(f), ptr(lins[ti]), ptr(val), B, co, f.shape[2] * f.shape[3],
                 0 if ti == 0 else 1, cd, ws.data_ptr(), ws.numel(), stream())
            ti += 1
        ctx.normalize, ctx.cd, ctx.B, ctx.in_shape, ctx.pred_dtype = int(normalize), cd, B, (B, 3, H, W), pred.dtype
        ctx.lins, ctx.net = lins, net
        ctx.save_for_backward(x, *feats, *wb)
        return val.view(B, 1, 1, 1)

    @staticmethod
    def backward(ctx, g):
        cfg = NETS[ctx.net]
        nl = len(cfg)
        saved = ctx.saved_tensors
        x, feats, wb = saved[0], saved[1:1 + nl], saved[1 + nl:]
        B, cd = ctx.B, ctx.cd
        gval = g.contiguous().view(B).float()
        grad = None                        # gradient w.r.t. feats[li][B:] (post-ReLU, gen half)
        ti = sum(1 for c in cfg if c[6]) - 1
        for li in reversed(range(nl)):
            ci, co, k, s, p, pool, tap, idx = cfg[li]
            f = feats[li]
            fgen = f[B:]
            if tap:
                if grad is None:
                    grad = torch.empty_like(fgen)
                    acc = 0
                else:
                    acc = 1
                call("hific_lpips_tap_bwd", ptr(f), ptr(ctx.lins[ti]), ptr(gval), ptr(grad), B, co,
                     f.shape[2] * f.shape[3], acc, cd, stream())
                ti -= 1
            dz = torch.empty_like(grad)
            call("hific_act_bwd", ptr(grad), ptr(fgen), ptr(dz), grad.numel(), 0.0, lib.dtype_code(grad), stream())
            # input of conv li: the pooled or plain feature map of layer li-1, or the prepped image
            prev_pool = cfg[li - 1][5] if li > 0 else None
            if li == 0:
                in_shape = (B,) + tuple(x.shape[1:])
            else:
                prev = feats[li - 1]
                if prev_pool:
                    in_shape = (B, prev.shape[1]) + _pooled_hw(prev_pool, prev.shape[2], prev.shape[3])
                else:
                    in_shape = (B,) + tuple(prev.shape[1:])
            din = _conv_bwd_data(dz, wb[2 * li], in_shape, s, p, cd)
            if prev_pool:
                prev_gen = feats[li - 1][B:]
                dprev = torch.empty_like(prev_gen)
                call("hific_maxpool3s2_bwd" if prev_pool == "3s2" else "hific_maxpool2s2_bwd", ptr(prev_gen), ptr(din),
                     ptr(dprev), B * prev_gen.shape[1], prev_gen.shape[2], prev_gen.shape[3], lib.dtype_code(prev_gen),
                     stream())
                grad = dprev
            else:
                grad = din
        # grad is now d/d(prepped gen image) -> d/d pred
        dpred = torch.empty(ctx.in_shape, dtype=ctx.pred_dtype, device=g.device)
        call("hific_lpips_prep_bwd", ptr(grad), ptr(dpred), B, ctx.in_shape[2] * ctx.in_shape[3],
             ctx.normalize, cd, 1 if ctx.pred_dtype == torch.float32 else 0, stream())
        return (dpred, None, None, None, None, None) + (None,) * len(wb)


class PerceptualLoss(nn.Module):
    """Like the reference (whose DistModel is not an nn.Module, base_model.py:5), the LPIPS tensors are *not*
    registered: they never appear in a HiFIC state_dict, but they follow `.to()/.cuda()` of the owning model."""

    def __init__(self, model='net-lin', net='alex', colorspace='rgb', spatial=False, use_gpu=True, gpu_ids=[0],
                 version='0.1', backbone_seed=1234, backbone_state_dict=None, backbone_path=None,
                 allow_random_backbone=False):
        """net: 'alex' (what HiFIC uses, src/model.py:101) or 'vgg' (VGG16, networks_basic.py:36-38, taps relu1_2 ..
        relu5_3).  Backbone weights, in order of precedence: `backbone_state_dict` (torchvision keys `features.N.*`),
        `backbone_path` / $HIFIC_LPIPS_ALEX_WEIGHTS resp. $HIFIC_LPIPS_VGG_WEIGHTS (a torch.save'd state_dict:
        torchvision's alexnet-owt-*.pth / vgg16-*.pth work as they are).  With
        neither, the backbone is a *seeded random* network - fine for benchmarks and parity tests
        (`allow_random_backbone=True` says so explicitly), wrong for real training: a loud warning is issued."""
        super().__init__()
        if model != 'net-lin' or net not in NETS or colorspace != 'rgb' or spatial or version != '0.1':
            raise NotImplementedError("hific_amd PerceptualLoss implements model='net-lin', net='alex' (HiFIC's choice) or "
                                      "'vgg', colorspace='rgb', spatial=False, version='0.1'")
        self.net = net
        cfg = NETS[net]
        self.use_gpu, self.gpu_ids, self.spatial = use_gpu, gpu_ids, spatial
        # backbone tensors in torchvision's `features.N.{weight,bias}` naming; frozen (requires_grad=False)
        gen = torch.Generator().manual_seed(backbone_seed)
        t = {}
        for (ci, co, k, s, p, _, _, idx) in cfg:
            bound = 1.0 / np.sqrt(ci * k * k)
            t[f"features.{idx}.weight"] = (torch.rand((co, ci, k, k), generator=gen) * 2 - 1) * bound
            t[f"features.{idx}.bias"] = (torch.rand((co,), generator=gen) * 2 - 1) * bound
        lin = np.load(os.path.join(_HERE, "weights", f"lpips_{net}_lin_v0.1.npz"))
        for i in range(5):
            t[f"lin{i}"] = torch.from_numpy(lin[f"lin{i}"].copy())
        object.__setattr__(self, "_t", t)          # plain dict: invisible to state_dict()/parameters()
        backbone_path = backbone_path or os.environ.get(f"HIFIC_LPIPS_{net.upper()}_WEIGHTS")
        self.backbone_source = "seeded-random"
        if backbone_state_dict is None and backbone_path:
            backbone_state_dict = torch.load(backbone_path, map_location="cpu")
            self.backbone_source = backbone_path
        elif backbone_state_dict is not None:
            self.backbone_source = "state_dict"
        if backbone_state_dict is not None:
            self.load_backbone_state_dict(backbone_state_dict)
        elif not allow_random_backbone:
            import logging
            import warnings
            msg = (f"hific_amd PerceptualLoss: NO pretrained {net} weights given - the LPIPS backbone is a seeded "
                   "RANDOM network (the reference downloads torchvision's ImageNet weights, pretrained_networks.py:59,99). "
                   "Training against it optimises a meaningless perceptual term. Pass backbone_path= / "
                   "backbone_state_dict=, set $HIFIC_LPIPS_ALEX_WEIGHTS / $HIFIC_LPIPS_VGG_WEIGHTS, or call load_backbone_state_dict(); "
                   "allow_random_backbone=True silences this for benchmarks and tests.")
            warnings.warn(msg, RuntimeWarning, stacklevel=2)
            logging.getLogger("hific_amd").warning(msg)
        if use_gpu and torch.cuda.is_available():
            dev = torch.device("cuda", gpu_ids[0] if gpu_ids else 0)
            for k in t:
                t[k] = t[k].to(dev)

    def _apply(self, fn, *a, **k):
        super()._apply(fn, *a, **k)
        for key in self._t:
            self._t[key] = fn(self._t[key])
        return self

    def __getattr__(self, name):
        if name.startswith("lin") and name[3:].isdigit():
            return self._t[name]
        return super().__getattr__(name)

    def load_backbone_state_dict(self, sd):
        """Accepts torchvision keys (`features.N.weight`) or bare `N.weight`."""
        self.backbone_source = "state_dict"
        with torch.no_grad():
            for idx in [c[7] for c in NETS[self.net]]:
                for nm in ("weight", "bias"):
                    key = f"features.{idx}.{nm}" if f"features.{idx}.{nm}" in sd else f"{idx}.{nm}"
                    dst = self._t[f"features.{idx}.{nm}"]
                    dst.copy_(sd[key].to(dst.device))

    def forward(self, pred, target, normalize=False):
        if target.requires_grad:
            raise NotImplementedError("gradient w.r.t. the LPIPS target image is not implemented (not on the HiFIC path)")
        lins = tuple(self._t[f"lin{i}"] for i in range(5))
        wb = []
        for idx in [c[7] for c in NETS[self.net]]:
            wb += [self._t[f"features.{idx}.weight"], self._t[f"features.{idx}.bias"]]
        pre = self.__dict__.pop("_prefetched", None)
        return LpipsFn.apply(pred.contiguous(), target.contiguous(), normalize, lins, pre, self.net, *wb)

    def prefetch_target(self, target, normalize=False):
        """Computes the target-image half of the feature maps now, on the current stream, for the next forward() with this
        very `target` (same tensor, unmodified).  A forward() with another target ignores it."""
        cdt = ops.get_compute_dtype()
        cd = lib.HIFIC_F32 if cdt == torch.float32 else lib.HIFIC_BF16
        target = target.contiguous()
        lib.require_gpu(target)
        wb = []
        cfg = NETS[self.net]
        for idx in [c[7] for c in cfg]:
            wb += [self._t[f"features.{idx}.weight"], self._t[f"features.{idx}.bias"]]
        B, _, H, W = target.shape
        with torch.no_grad():
            xt = torch.empty((2 * B, 3, H, W), dtype=cdt, device=target.device)
            tf = 1 if target.dtype == torch.float32 else 0
            call("hific_lpips_prep", ptr(target), tf, ptr(target), tf, ptr(xt), B, H * W, int(normalize), cd, stream())
            feats = [None] * len(cfg)
            _net_half(xt[:B], wb, cd, feats, 0, B, cfg)
        ev = torch.cuda.current_stream(target.device).record_event()
        self.__dict__["_prefetched"] = TargetFeatures(_target_key(target, normalize, cd, self.net), feats, ev, target)

    def drop_prefetch(self):
        """Forget a prefetch nobody consumed (Model.forward calls this first: a forward that raised must not leave one)."""
        self.__dict__.pop("_prefetched", None)

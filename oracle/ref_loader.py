"""ORACLE SUPPORT — TEST INFRASTRUCTURE ONLY.  Imports the *real* reference (read-only, /root/reference) in this
container so the restatement in hific_oracle.py can be pinned against it and golden vectors can be generated.
The GPU box has no /root/reference: nothing at run time there may depend on this file (`available()` is False).

Three packages the reference imports are not installed and cannot be (no network): torchvision, skimage, autograd.
They are replaced by minimal import shims:
  * torchvision.models.alexnet(...).features — the standard AlexNet feature stack (indices matter for the slicing
    at pretrained_networks.py:66-75) with seeded random weights (ImageNet weights are unobtainable offline)
  * skimage / autograd — empty stubs (never called on the hot path)
"""
import os
import sys
import types

REF_ROOT = os.environ.get("HIFIC_REFERENCE", "/root/reference")


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "src"))


class _Permissive(types.ModuleType):
    """Module whose unknown attributes resolve to an inert placeholder class (usable as a base class)."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        cls = type(name, (object,), {"__init__": lambda self, *a, **k: None})
        setattr(self, name, cls)
        return cls


def _install_shims():
    import torch
    import torch.nn as nn

    if "torchvision" not in sys.modules:
        tv = _Permissive("torchvision")
        models = _Permissive("torchvision.models")
        utils = _Permissive("torchvision.utils")
        transforms = _Permissive("torchvision.transforms")
        datasets = _Permissive("torchvision.datasets")

        class _AlexNet(nn.Module):
            def __init__(self):
                super().__init__()
                self.features = nn.Sequential(
                    nn.Conv2d(3, 64, kernel_size=11, stride=4, padding=2), nn.ReLU(inplace=True),
                    nn.MaxPool2d(kernel_size=3, stride=2),
                    nn.Conv2d(64, 192, kernel_size=5, padding=2), nn.ReLU(inplace=True),
                    nn.MaxPool2d(kernel_size=3, stride=2),
                    nn.Conv2d(192, 384, kernel_size=3, padding=1), nn.ReLU(inplace=True),
                    nn.Conv2d(384, 256, kernel_size=3, padding=1), nn.ReLU(inplace=True),
                    nn.Conv2d(256, 256, kernel_size=3, padding=1), nn.ReLU(inplace=True),
                    nn.MaxPool2d(kernel_size=3, stride=2),
                )

        def alexnet(pretrained=False, **kw):
            return _AlexNet()

        class _VGG16(nn.Module):
            def __init__(self):
                super().__init__()
                layers, c = [], 3
                for v in (64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"):
                    if v == "M":
                        layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
                    else:
                        layers += [nn.Conv2d(c, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
                        c = v
                self.features = nn.Sequential(*layers)

        def vgg16(pretrained=False, **kw):
            return _VGG16()

        def _unavailable(*a, **k):
            raise RuntimeError("torchvision shim: only alexnet() and vgg16() are provided")

        models.alexnet = alexnet
        models.vgg16 = vgg16
        models.squeezenet1_1 = _unavailable
        models.resnet18 = models.resnet34 = models.resnet50 = models.resnet101 = models.resnet152 = _unavailable
        utils.save_image = _unavailable
        utils.make_grid = _unavailable
        transforms.Compose = transforms.ToTensor = transforms.Normalize = _unavailable
        datasets.ImageFolder = object
        tv.models, tv.utils, tv.transforms, tv.datasets = models, utils, transforms, datasets
        for n, m in (("torchvision", tv), ("torchvision.models", models), ("torchvision.utils", utils),
                     ("torchvision.transforms", transforms), ("torchvision.datasets", datasets)):
            sys.modules[n] = m
    if "skimage" not in sys.modules:
        sk = _Permissive("skimage")
        for sub in ("measure", "io", "color", "transform"):
            m = _Permissive("skimage." + sub)
            setattr(sk, sub, m)
            sys.modules["skimage." + sub] = m
        sk.measure.compare_ssim = None
        sk.io.imread = None
        sys.modules["skimage"] = sk
    if "autograd" not in sys.modules:
        import numpy
        ag = types.ModuleType("autograd")
        ag.make_vjp = None
        ext = types.ModuleType("autograd.extend")
        ext.vspace = None
        ext.VSpace = object
        agn = types.ModuleType("autograd.numpy")
        agn.__dict__.update({k: getattr(numpy, k) for k in dir(numpy) if not k.startswith("__")})
        ag.extend, ag.numpy = ext, agn
        sys.modules["autograd"], sys.modules["autograd.extend"], sys.modules["autograd.numpy"] = ag, ext, agn


def load(root=None):
    """Returns the reference's top-level modules as a namespace: .model, .default_config, .hyperprior, ...
    `root`: directory holding `default_config.py` and `src/` (default /root/reference; the GPU test passes the
    directory it unpacked oracle/_ref/reference_src.tar.gz into)."""
    global REF_ROOT
    if root is not None:
        REF_ROOT = root
    if not available():
        raise RuntimeError("reference not present (expected at %s)" % REF_ROOT)
    _install_shims()
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    import importlib
    ns = types.SimpleNamespace()
    ns.default_config = importlib.import_module("default_config")
    ns.encoder = importlib.import_module("src.network.encoder")
    ns.generator = importlib.import_module("src.network.generator")
    ns.discriminator = importlib.import_module("src.network.discriminator")
    ns.hyper = importlib.import_module("src.network.hyper")
    ns.channel = importlib.import_module("src.normalisation.channel")
    ns.hyperprior = importlib.import_module("src.hyperprior")
    ns.hyperprior_model = importlib.import_module("src.compression.hyperprior_model")
    ns.maths = importlib.import_module("src.helpers.maths")
    ns.utils = importlib.import_module("src.helpers.utils")
    ns.losses = importlib.import_module("src.loss.losses")
    ns.perceptual_loss = importlib.import_module("src.loss.perceptual_similarity.perceptual_loss")
    ns.model = importlib.import_module("src.model")
    return ns


def build_reference_model(ns, gan=False, training=True, log_dir="/tmp/hific_ref_logs", **overrides):
    """Reference src.model.Model on CPU (TRAINING mode), with a throw-away logger."""
    import logging
    cfg = ns.default_config
    base = cfg.hific_args if gan else cfg.mse_lpips_args
    d = {}
    for klass in reversed(base.__mro__):
        d.update({k: v for k, v in vars(klass).items() if not k.startswith("__")})
    d.update(overrides)
    args = ns.utils.Struct(**d)
    logger = logging.getLogger("hific_ref")
    mtype = cfg.ModelTypes.COMPRESSION_GAN if gan else cfg.ModelTypes.COMPRESSION
    m = ns.model.Model(args, logger, model_mode=cfg.ModelModes.TRAINING, model_type=mtype)
    m.train(training)
    return m


def set_lpips_backbone(ref_model, backbone_sd):
    """Load a seeded backbone (oracle.make_alex_backbone / make_vgg_backbone) into the reference's LPIPS net;
    `ref_model` is the reference Model or a reference PerceptualLoss."""
    import torch
    pl = getattr(ref_model, "perceptual_loss", ref_model)
    net = pl.model.net   # PNetLin
    feats = {}
    for sl in (net.net.slice1, net.net.slice2, net.net.slice3, net.net.slice4, net.net.slice5):
        for name, mod in sl.named_children():
            feats[name] = mod
    with torch.no_grad():
        for k, v in backbone_sd.items():
            _, idx, nm = k.split(".")
            getattr(feats[idx], nm).copy_(v)


def reference_lins(ref_model):
    net = getattr(ref_model, "perceptual_loss", ref_model).model.net
    return [l.model[-1].weight.detach().reshape(-1).clone() for l in net.lins]

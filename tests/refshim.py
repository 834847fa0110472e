"""Import shims that let the READ-ONLY reference checkout run in this container
(torch 2.10, no timm/easydict).  Used only by tests and by tests/golden/make_golden.py;
nothing is written to /root/reference and nothing here is product code.

  * torch._six.container_abcs  (removed in torch 2.x; AutoFormer/model/utils.py:5)
  * easydict.EasyDict          (iRPE/DeiT-with-iRPE/irpe.py:2)
  * the handful of timm symbols rpe_vision_transformer.py / models.py / rpe_models.py import
"""
import collections.abc
import importlib
import os
import sys
import types

import torch
import torch.nn as nn

REFERENCE = "/root/reference"
AUTOFORMER = os.path.join(REFERENCE, "AutoFormer")
IRPE = os.path.join(REFERENCE, "iRPE", "DeiT-with-iRPE")
MINIVIT = os.path.join(REFERENCE, "MiniViT", "Mini-DeiT")
DETR_RPE = os.path.join(REFERENCE, "iRPE", "DETR-with-iRPE", "models", "rpe_attention")
OPEN_CLIP = os.path.join(REFERENCE, "TinyCLIP", "src", "open_clip")


def have_reference():
    return os.path.isdir(REFERENCE)


def _install_torch_six():
    if "torch._six" not in sys.modules:
        m = types.ModuleType("torch._six")
        m.container_abcs = collections.abc
        m.string_classes = (str, bytes)
        m.int_classes = (int,)
        sys.modules["torch._six"] = m
        torch._six = m


class _EasyDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _install_easydict():
    if "easydict" not in sys.modules:
        m = types.ModuleType("easydict")
        m.EasyDict = _EasyDict
        sys.modules["easydict"] = m


def _install_timm_stub():
    """Just enough of timm 0.3.2 for the iRPE DeiT model files to import and build."""
    if "timm" in sys.modules:
        return
    from itertools import repeat

    def to_2tuple(x):
        return x if isinstance(x, collections.abc.Iterable) else tuple(repeat(x, 2))

    def trunc_normal_(tensor, mean=0., std=1., a=-2., b=2.):
        return nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)

    class DropPath(nn.Module):
        def __init__(self, drop_prob=None):
            super().__init__()
            self.drop_prob = drop_prob

        def forward(self, x):
            if not self.drop_prob or not self.training:
                return x
            keep = 1 - self.drop_prob
            shape = (x.shape[0],) + (1,) * (x.ndim - 1)
            mask = (keep + torch.rand(shape, dtype=x.dtype, device=x.device)).floor_()
            return x.div(keep) * mask

    class Mlp(nn.Module):
        def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
            super().__init__()
            out_features = out_features or in_features
            hidden_features = hidden_features or in_features
            self.fc1 = nn.Linear(in_features, hidden_features)
            self.act = act_layer()
            self.fc2 = nn.Linear(hidden_features, out_features)
            self.drop = nn.Dropout(drop)

        def forward(self, x):
            return self.drop(self.fc2(self.drop(self.act(self.fc1(x)))))

    class PatchEmbed(nn.Module):
        def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
            super().__init__()
            img_size, patch_size = to_2tuple(img_size), to_2tuple(patch_size)
            self.img_size, self.patch_size = img_size, patch_size
            self.num_patches = (img_size[1] // patch_size[1]) * (img_size[0] // patch_size[0])
            self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)

        def forward(self, x):
            return self.proj(x).flatten(2).transpose(1, 2)

    class HybridEmbed(nn.Module):
        def __init__(self, *a, **k):
            raise NotImplementedError("HybridEmbed is not part of the hot path")

    def _cfg(url="", **kwargs):
        return dict(url=url, num_classes=1000, input_size=(3, 224, 224), **kwargs)

    def register_model(fn):
        return fn

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    timm = mod("timm")
    timm.data = mod("timm.data", IMAGENET_DEFAULT_MEAN=(0.485, 0.456, 0.406),
                    IMAGENET_DEFAULT_STD=(0.229, 0.224, 0.225))
    timm.models = mod("timm.models")
    timm.models.helpers = mod("timm.models.helpers", load_pretrained=lambda *a, **k: None)
    timm.models.layers = mod("timm.models.layers", DropPath=DropPath, to_2tuple=to_2tuple,
                             trunc_normal_=trunc_normal_)
    timm.models.resnet = mod("timm.models.resnet", resnet26d=None, resnet50d=None)
    timm.models.registry = mod("timm.models.registry", register_model=register_model)
    timm.models.vision_transformer = mod("timm.models.vision_transformer", _cfg=_cfg, default_cfgs={},
                                         Mlp=Mlp, PatchEmbed=PatchEmbed, HybridEmbed=HybridEmbed)


def _purge(prefixes):
    for k in list(sys.modules):
        if any(k == p or k.startswith(p + ".") for p in prefixes):
            del sys.modules[k]


class _Path:
    def __init__(self, path, purge):
        self.path, self.purge = path, purge

    def __enter__(self):
        _purge(self.purge)
        sys.path.insert(0, self.path)

    def __exit__(self, *exc):
        sys.path.remove(self.path)
        _purge(self.purge)


def load_autoformer_reference():
    """-> namespace with the reference's Vision_TransformerSuper, module classes and
    sample_configs (supernet_engine.py:13-24 needs timm -> restated import-free)."""
    _install_torch_six()
    with _Path(AUTOFORMER, ["model"]):
        st = importlib.import_module("model.supernet_transformer")
        mh = importlib.import_module("model.module.multihead_super")
        ns = types.SimpleNamespace(
            Vision_TransformerSuper=st.Vision_TransformerSuper,
            TransformerEncoderLayer=st.TransformerEncoderLayer,
            AttentionSuper=mh.AttentionSuper,
            RelativePosition2D_super=mh.RelativePosition2D_super,
            LinearSuper=importlib.import_module("model.module.Linear_super").LinearSuper,
            qkv_super=importlib.import_module("model.module.qkv_super").qkv_super,
            LayerNormSuper=importlib.import_module("model.module.layernorm_super").LayerNormSuper,
            PatchembedSuper=importlib.import_module("model.module.embedding_super").PatchembedSuper,
        )
    return ns


def reference_sample_configs():
    """Extract the reference's sample_configs without importing timm: exec only that
    function's source from supernet_engine.py (read-only)."""
    import ast
    import random
    src = open(os.path.join(AUTOFORMER, "supernet_engine.py")).read()
    tree = ast.parse(src)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "sample_configs"][0]
    ns = {"random": random}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "supernet_engine.py", "exec"), ns)
    return ns["sample_configs"]


def load_irpe_reference(with_dropin=False):
    """-> the reference's irpe module (pure-PyTorch fallback unless with_dropin)."""
    _install_easydict()
    purge = ["irpe", "rpe_ops", "rpe_index_cpp"]
    _purge(purge)
    if with_dropin:
        import cream_amd.dropin as d
        d.install()
    else:
        # make sure `import rpe_ops` fails so that irpe.py takes its fallback path
        import cream_amd.dropin as d
        if d.PATH in sys.path:
            sys.path.remove(d.PATH)
    sys.path.insert(0, IRPE)
    try:
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            irpe = importlib.import_module("irpe")
    finally:
        sys.path.remove(IRPE)
    return irpe


def load_irpe_models():
    """-> (irpe, rpe_vision_transformer, models, rpe_models) of DeiT-with-iRPE, fallback path."""
    _install_easydict()
    _install_timm_stub()
    _purge(["irpe", "rpe_ops", "rpe_index_cpp", "rpe_vision_transformer", "models", "rpe_models"])
    import cream_amd.dropin as d
    if d.PATH in sys.path:
        sys.path.remove(d.PATH)
    sys.path.insert(0, IRPE)
    try:
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            mods = [importlib.import_module(m) for m in ("irpe", "rpe_vision_transformer", "models", "rpe_models")]
    finally:
        sys.path.remove(IRPE)
    return mods


def load_minivit_models():
    """-> (irpe, mini_vision_transformer, models, mini_deit_models) of MiniViT/Mini-DeiT, fallback rpe path."""
    _install_easydict()
    _install_timm_stub()
    names = ("irpe", "mini_vision_transformer", "models", "mini_deit_models")
    _purge(["rpe_ops", "rpe_index_cpp", "rpe_vision_transformer", "rpe_models"] + list(names))
    import cream_amd.dropin as d
    if d.PATH in sys.path:
        sys.path.remove(d.PATH)
    sys.path.insert(0, MINIVIT)
    try:
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            mods = [importlib.import_module(m) for m in names]
    finally:
        sys.path.remove(MINIVIT)
        _purge(names)
    return mods


def load_detr_rpe_attention():
    """-> (irpe, multi_head_attention) of DETR-with-iRPE/models/rpe_attention, imported as a stand-alone package (the
    parent `models/__init__.py` pulls the whole detector in), fallback rpe path."""
    _install_easydict()
    name = "_ref_detr_rpe_attention"
    _purge([name, "rpe_ops", "rpe_index_cpp"])
    import cream_amd.dropin as d
    if d.PATH in sys.path:
        sys.path.remove(d.PATH)
    pkg = types.ModuleType(name)
    pkg.__path__ = [DETR_RPE]
    sys.modules[name] = pkg
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mha = importlib.import_module(name + ".multi_head_attention")
        irpe = importlib.import_module(name + ".irpe")
    return irpe, mha


def load_tinyclip_model():
    """-> TinyCLIP/src/open_clip/model.py imported as a stand-alone package module.  Its imports that are absent here
    (torchvision for a frozen-BatchNorm helper of the ResNet tower, timm for the timm tower) get empty stand-ins: neither
    is touched by the ViT towers."""
    name = "_ref_open_clip"
    _purge([name])
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tv.ops = types.ModuleType("torchvision.ops")
        tv.ops.misc = types.ModuleType("torchvision.ops.misc")
        tv.ops.misc.FrozenBatchNorm2d = type("FrozenBatchNorm2d", (nn.Module,), {})
        sys.modules.update({"torchvision": tv, "torchvision.ops": tv.ops, "torchvision.ops.misc": tv.ops.misc})
    pkg = types.ModuleType(name)
    pkg.__path__ = [OPEN_CLIP]
    sys.modules[name] = pkg
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return importlib.import_module(name + ".model")

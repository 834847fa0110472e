"""Import-path drop-ins: modules named exactly as the reference's callers import them
(`rpe_index_cpp`, `rpe_ops.rpe_index`, `model.module.*`, `model.utils`).

    import cream_amd.dropin as d; d.install()

prepends this directory to sys.path so that the reference's unchanged callers
(`irpe.py`, `rpe_vision_transformer.py`, `model/supernet_transformer.py`) resolve their
imports to the MI355X implementations.  See INTEGRATION.md.
"""
import os
import sys

PATH = os.path.dirname(os.path.abspath(__file__))


def install():
    if PATH not in sys.path:
        sys.path.insert(0, PATH)
    return PATH


def install_autoformer(reference_model_dir=None, fast_path=True):
    """Make `import model.supernet_transformer` resolve the REFERENCE's unchanged
    supernet_transformer.py (from `reference_model_dir`, e.g. .../AutoFormer/model) while
    `model.module.*` and `model.utils` resolve to the MI355X implementations here.  With
    `fast_path` the caller's classes are patched at import (SURVEY 8b allows exactly this) so that a
    bf16-autocast forward on the GPU runs the whole block stack as one native autograd node
    (`enable_fast_path`); everything else keeps executing the reference's own code."""
    import importlib
    install()
    for k in [k for k in sys.modules if k == "model" or k.startswith("model.")]:
        del sys.modules[k]
    pkg = importlib.import_module("model")
    if reference_model_dir and reference_model_dir not in pkg.__path__:
        pkg.__path__.append(reference_model_dir)      # searched AFTER our directory
        if fast_path:
            enable_fast_path(importlib.import_module("model.supernet_transformer"))
    return pkg


def enable_fast_path(caller):
    """Patch a caller module that defines `Vision_TransformerSuper` / `TransformerEncoderLayer` with
    the reference's attribute names (AutoFormer/model/supernet_transformer.py:20-172, 175-294): its
    `forward_features` is replaced by the one of cream_amd/autoformer/supernet.py, which hands the run
    of active blocks to `block.StackFunction` (one native call per block and direction) when the
    input is a CUDA tensor under bf16 autocast and every block is supported — and otherwise runs the
    caller's OWN `TransformerEncoderLayer.forward` block by block, i.e. the unchanged file."""
    from cream_amd.autoformer import supernet as ours
    layer, model = caller.TransformerEncoderLayer, caller.Vision_TransformerSuper
    if getattr(model, "_cream_fast_path", False):
        return caller
    layer.fused = True
    layer._dp = None
    layer.drop_path_scales = ours.TransformerEncoderLayer.drop_path_scales
    model._keep_prob = ours.Vision_TransformerSuper._keep_prob
    model._cream_reference_forward_features = model.forward_features
    model.forward_features = ours.Vision_TransformerSuper.forward_features
    model._cream_fast_path = True
    return caller


def enable_irpe_fast_path(caller):
    """Patch a caller module that defines `RPEAttention` with the reference's attribute names
    (iRPE/DeiT-with-iRPE/rpe_vision_transformer.py:45-97: qkv, proj, attn_drop, proj_drop, rpe_q / rpe_k / rpe_v,
    scale, num_heads): its `forward` is replaced by cream_amd.rpe_attention.RPEAttention.forward, which runs the
    fused kernels of csrc/irpe_attn.hip (`cream_irpe_attn_fwd / _bwd`) when the layer is covered — CUDA, bf16
    operands, head_dim 64, contextual rpe modules built from the drop-in `irpe` (<= 64 buckets) — and otherwise
    computes exactly what the reference's forward computes, on the drop-in `rpe_index` operator."""
    from cream_amd.rpe_attention import RPEAttention as ours, VisionTransformer as ours_model
    cls = caller.RPEAttention
    if not getattr(cls, "_cream_fast_path", False):
        cls._cream_reference_forward = cls.forward
        cls.forward = ours.forward
        cls._cream_fast_path = True
    # the model's block loop (rpe_vision_transformer.py:183-196): under bf16 autocast the whole run of RPEBlocks becomes one node on
    # the own kernels (cream_amd/deit_native.py: same attribute names — norm1 / attn.qkv / attn.proj / norm2 / mlp.fc1 / mlp.fc2 /
    # drop_path); everything that node does not cover runs the caller's own blocks one by one, as before
    model = getattr(caller, "VisionTransformer", None)
    if model is not None and not getattr(model, "_cream_fast_path", False):
        model._cream_reference_forward_features = model.forward_features
        model.forward_features = ours_model.forward_features
        model._cream_fast_path = True
    return caller


def install_irpe(reference_dir=None, fast_path=True):
    """`import irpe`, `import rpe_ops.rpe_index`, `import rpe_index_cpp` resolve to the MI355X implementations
    (install()); with `reference_dir` (e.g. .../iRPE/DeiT-with-iRPE) the reference's unchanged
    `rpe_vision_transformer.py` is imported from there and — with `fast_path` — its RPEAttention patched
    (enable_irpe_fast_path).  Returns the caller module, or None without a reference directory."""
    import importlib
    install()
    if not reference_dir:
        return None
    for k in [k for k in sys.modules if k in ("irpe", "rpe_vision_transformer", "rpe_index_cpp") or k.startswith("rpe_ops")]:
        del sys.modules[k]
    if reference_dir not in sys.path:
        sys.path.append(reference_dir)                # AFTER the drop-in directory: `irpe` stays ours
    caller = importlib.import_module("rpe_vision_transformer")
    return enable_irpe_fast_path(caller) if fast_path else caller

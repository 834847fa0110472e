"""DeiT-with-iRPE blocks on the framework's own kernels — the run of `RPEBlock`s (iRPE/DeiT-with-iRPE/rpe_vision_transformer.py:100-117,
:193-199) of a `VisionTransformer` as ONE autograd node under bf16 autocast:

    x1 = x  + proj(attn(norm1(x)))          attn = RPEAttention's core with rpe on any subset of q / k / v (csrc/irpe_attn.hip)
    x2 = x1 + fc2(gelu(fc1(norm2(x1))))

The node is cream_amd.tinyclip.native.TowerStack (LayerNorm passes with the residual add folded in, own MFMA GEMMs with bias /
erf-GELU / GELU' epilogues on bf16 operand copies, split-K weight gradients on a side stream, fp32 residual stream); this file only
says where an RPEBlock keeps its parameters (`BlockView`) and when the node applies.  The lookup tables of the rpe terms get their
gradients from the attention backward (cream_irpe_table_grad) and are accumulated into `.grad` like every other parameter of the node.

Applies to: CUDA tensors under bf16 autocast, heads of 64, exact-erf GELU, no projection / MLP dropout active (attention dropout is
fine: in-kernel; stochastic depth too: per-sample factors of the two branches through the residual / LayerNorm kernels), every rpe either absent or a plain iRPE the fused kernels take (<= 64 buckets, contextual or
bias mode; the cross method keeps the module path: its one-table view is differentiated by autograd), all parameters trainable (or
no gradient wanted at all).  `CREAM_DEIT_NATIVE=0` keeps the module path.
"""
import os

import torch
import torch.nn as nn

from . import irpe_fused
from .irpe import iRPE
from .tinyclip import native


def view_of(blk):
    at, mlp = blk.attn, blk.mlp
    return native.BlockView(blk.norm1, blk.norm2, [(at.qkv.weight, at.qkv.bias), (at.proj.weight, at.proj.bias),
                                                   (mlp.fc1.weight, mlp.fc1.bias), (mlp.fc2.weight, mlp.fc2.bias)],
                            at.num_heads, (at.rpe_q, at.rpe_k, at.rpe_v), float(at.attn_drop.p), float(at.scale),
                            float(getattr(blk.drop_path, "drop_prob", 0.0) or 0.0))


def _inactive(mod, training):
    """An nn.Dropout / DropPath / Identity that does nothing in this call."""
    if isinstance(mod, nn.Identity):
        return True
    p = getattr(mod, "p", getattr(mod, "drop_prob", None))
    return p is not None and (not training or not p)


def _block_supported(blk, L, dev):
    at, mlp = blk.attn, blk.mlp
    D = at.qkv.weight.shape[1]
    if at.qkv.bias is None or D != at.num_heads * 64 or mlp.fc1.weight.shape[0] % 8 or D % 8:
        return False
    if not (isinstance(mlp.act, nn.GELU) and getattr(mlp.act, "approximate", "none") == "none"):
        return False
    if not (isinstance(blk.norm1, nn.LayerNorm) and isinstance(blk.norm2, nn.LayerNorm) and blk.norm1.elementwise_affine):
        return False
    tr = blk.training
    if not (_inactive(at.proj_drop, tr) and _inactive(mlp.drop, tr)):
        return False
    if not (isinstance(blk.drop_path, nn.Identity) or hasattr(blk.drop_path, "drop_prob")):     # DropPath: per-sample factors in the node
        return False
    rpes = (at.rpe_q, at.rpe_k, at.rpe_v)
    if any(r is not None and type(r) is not iRPE for r in rpes):
        return False
    p = float(at.attn_drop.p) if tr else 0.0
    return at.qkv.weight.dtype == torch.float32 and irpe_fused.usable(torch.bfloat16, dev, 64, L, rpes, dropout_p=p)


def supported(blocks, x):
    if os.environ.get("CREAM_DEIT_NATIVE", "1") == "0" or not len(blocks):
        return False
    if not (x.is_cuda and x.dim() == 3 and torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16
            and x.shape[1] <= 2048):
        return False
    if not all(_block_supported(blk, x.shape[1], x.device) for blk in blocks):
        return False
    if torch.is_grad_enabled():
        flags = [p.requires_grad for blk in blocks for p in blk.parameters()]
        if (x.requires_grad or any(flags)) and not all(flags):
            return False
    return True


def run(blocks, x):
    """x (B, L, D) -> (B, L, D): all blocks in one node."""
    return native.stack(blocks, view_of, x)

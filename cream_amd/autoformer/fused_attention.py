"""Fused attention + 2-D relative position bias on the HIP kernels cream_attn_rpe2d_fwd /
cream_attn_rpe2d_bwd (cream_amd/csrc/attn_rpe2d.hip) — the core of AttentionSuper.forward
between the qkv and proj GEMMs (AutoFormer/model/module/multihead_super.py:135-154).

One launch per direction for all (batch, head) pairs; nothing of size N^2 reaches HBM; the
relative position index matrices are generated in-kernel from the grid geometry.  Attention
dropout (multihead_super.py:145) runs in-kernel too: the keep mask is a counter-based function
of (seed, b, h, i, j) that the backward regenerates (cream_attn_rpe2d_fwd_drop / _bwd_drop).  There
is no fallback here: if the geometry is outside the kernel family, `supported()` says so
and the caller picks the bucketed HIP path (attention_op) explicitly.
"""
import ctypes

import torch

from .. import _lib, timing

_DT = {torch.bfloat16: _lib.BF16, torch.float32: _lib.F32}


def padded_len(n):
    return (n + 31) // 32 * 32


def grid_of(n, max_relative_position):
    """(gh, gw) of a length-n sequence as the reference derives it (multihead_super.py:47:
    side = int((n-1)**0.5)), or None when the fused kernel family does not cover it."""
    side = int((n - 1) ** 0.5)
    if side < 1 or side * side != n - 1:
        return None
    if n > 256 or 2 * side + 1 > 32 or 2 * max_relative_position + 2 > 32:
        return None
    return side, side


def available():
    return True


def supported(qkv, dropout_p, max_relative_position=14, tables=()):
    if not (0.0 <= dropout_p < 1.0) or not qkv.is_cuda or qkv.dtype not in _DT:
        return False
    B, N, three, H, D = qkv.shape
    if D != 64 or grid_of(N, max_relative_position) is None:
        return False
    if qkv.stride(4) != 1 or any((s * qkv.element_size()) % 16 for s in qkv.stride()[:4]):
        return False
    for t in tables:
        if t.dtype != torch.float32 or t.stride(1) != 1 or t.shape[1] != 64 or t.stride(0) % 4:
            return False
    return True


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _flops(B, H, N):
    """ALGORITHMIC forward flops of the attention core: QK^T and PV at 2*N*N*64
    each per head, plus the bucketed relative-position terms (two 30-row tables on the key and
    on the value side: 4*N*64*60).  Padding, the one-hot extension columns and recomputation
    are NOT counted.  Backward is taken as 2.5x forward (dV, dP, dQ, dK + one score recompute)."""
    return B * H * (4 * N * N * 64 + 4 * N * 64 * 60)


def table_images(tkv, tkh, tvv, tvh, mr):
    """bf16 operand images of the four tables (cream_attn_rpe2d_table_images): one small launch.  The block path keeps
    them current through the optimizer kernel instead (block.BlockOperands.timg)."""
    lib = _lib.load()
    img = torch.empty((lib.cream_attn_rpe2d_table_image_bytes() // 2,), dtype=torch.bfloat16, device=tkv.device)
    with torch.cuda.device(tkv.device):
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(lib.cream_attn_rpe2d_table_images(_ptr(img), _ptr(tkv), _ptr(tkh), _ptr(tvv), _ptr(tvh), tkv.stride(0), mr, st),
                   "cream_attn_rpe2d_table_images")
    return img


def _wants_images(qkv, N, mr):
    return qkv.dtype == torch.bfloat16 and N == 197 and mr == 14


def attn_fwd_raw(qkv, tkv, tkh, tvv, tvh, scale, mr, timg=None, drop_p=0.0, seed=0):
    """qkv (B, N, 3, H, 64) -> (out (B, N, H, 64), lse (B, H, N), sp (B, H, 64, NP)).  One launch (plus the image
    launch when the AutoFormer geometry is asked for in bf16 without images: the ping-pong kernel needs them).
    drop_p > 0: attention dropout under the keep mask of `seed` (irpe_fused.dropout_keep_mask restates it)."""
    B, N, _, H, D = qkv.shape
    if timg is None and drop_p == 0.0 and _wants_images(qkv, N, mr):
        timg = table_images(tkv, tkh, tvv, tvh, mr)
    gh, gw = grid_of(N, mr)
    NP = padded_len(N)
    out = torch.empty((B, N, H, D), dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty((B, H, N), dtype=torch.float32, device=qkv.device)
    sp = torch.empty((B, H, 64, NP), dtype=qkv.dtype, device=qkv.device)
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    sb, sn, sh = qkv.stride(0), qkv.stride(1), qkv.stride(3)
    lib = _lib.load()
    with torch.cuda.device(qkv.device), timing.region("attn_rpe2d_fwd", flops=_flops(B, H, N)):
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(lib.cream_attn_rpe2d_fwd_drop(
            _ptr(out), _ptr(lse), _ptr(sp), _ptr(q), _ptr(k), _ptr(v), sb, sn, sh,
            _ptr(tkv), _ptr(tkh), _ptr(tvv), _ptr(tvh), tkv.stride(0), _ptr(timg) if timg is not None else None,
            B, H, N, gh, gw, mr, float(scale), float(drop_p), int(seed) & 0xFFFFFFFF, _DT[qkv.dtype], st),
            "cream_attn_rpe2d_fwd_drop")
    return out, lse, sp


def attn_bwd_raw(dout, qkv, tkv, tkh, tvv, tvh, out, lse, sp, scale, mr, reduce_tables=True, timg=None, drop_p=0.0, seed=0):
    """-> (dqkv (B, N, 3, H, 64), dtab (4, 32, 64) fp32 = gradients of [tkv, tkh, tvv, tvh] rows;
    with reduce_tables=False the per-workgroup partials (cream_attn_rpe2d_dtab_parts(B, H), 4, 32, 64) for
    cream_grad_finalize)."""
    B, N, _, H, D = qkv.shape
    gh, gw = grid_of(N, mr)
    NP = padded_len(N)
    dout = dout.contiguous()
    dqkv = torch.empty((B, N, 3, H, D), dtype=qkv.dtype, device=qkv.device)
    # side buffers of the two backward launches (see include/cream_amd.h)
    dlt = torch.empty((B, H, 64, NP), dtype=qkv.dtype, device=qkv.device)
    qe = torch.empty((B, H, NP, 32), dtype=qkv.dtype, device=qkv.device)
    de = torch.empty((B, H, NP, 32), dtype=qkv.dtype, device=qkv.device)
    delta = torch.empty((B, H, NP), dtype=torch.float32, device=qkv.device)
    dtab = torch.empty((_lib.load().cream_attn_rpe2d_dtab_parts(B, H), 4, 32, 64), dtype=torch.float32, device=qkv.device)
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    sb, sn, sh = qkv.stride(0), qkv.stride(1), qkv.stride(3)
    dq, dk, dv = dqkv[:, :, 0], dqkv[:, :, 1], dqkv[:, :, 2]
    dsb, dsn, dsh = dqkv.stride(0), dqkv.stride(1), dqkv.stride(3)
    lib = _lib.load()
    with torch.cuda.device(qkv.device), timing.region("attn_rpe2d_bwd", flops=int(2.5 * _flops(B, H, N))):
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(lib.cream_attn_rpe2d_bwd_drop(
            _ptr(dq), _ptr(dk), _ptr(dv), dsb, dsn, dsh, _ptr(dtab),
            _ptr(dlt), _ptr(qe), _ptr(de), _ptr(delta),
            _ptr(dout), _ptr(out), _ptr(lse), _ptr(sp), _ptr(q), _ptr(k), _ptr(v), sb, sn, sh,
            _ptr(tkv), _ptr(tkh), _ptr(tvv), _ptr(tvh), tkv.stride(0), _ptr(timg) if timg is not None else None,
            B, H, N, gh, gw, mr, float(scale), float(drop_p), int(seed) & 0xFFFFFFFF, _DT[qkv.dtype], st),
            "cream_attn_rpe2d_bwd_drop")
    if not reduce_tables:
        return dqkv, dtab
    return dqkv, dtab.sum(dim=0)                              # fixed-order reduction over the workgroups' partials


class _FusedAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, tkv, tkh, tvv, tvh, scale, mr, drop_p=0.0, seed=0):
        B, N = qkv.shape[:2]
        timg = table_images(tkv, tkh, tvv, tvh, mr) if drop_p == 0.0 and _wants_images(qkv, N, mr) else None
        out, lse, sp = attn_fwd_raw(qkv, tkv, tkh, tvv, tvh, scale, mr, timg=timg, drop_p=drop_p, seed=seed)
        ctx.save_for_backward(qkv, tkv, tkh, tvv, tvh, out, lse, sp)
        ctx.scale, ctx.mr, ctx.timg, ctx.drop = float(scale), mr, timg, (float(drop_p), int(seed))
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, tkv, tkh, tvv, tvh, out, lse, sp = ctx.saved_tensors
        dqkv, dt = attn_bwd_raw(dout, qkv, tkv, tkh, tvv, tvh, out, lse, sp, ctx.scale, ctx.mr, timg=ctx.timg,
                                drop_p=ctx.drop[0], seed=ctx.drop[1])
        nb = tkv.shape[0]
        return (dqkv, dt[0, :nb].to(tkv.dtype), dt[1, :nb].to(tkh.dtype), dt[2, :nb].to(tvv.dtype),
                dt[3, :nb].to(tvh.dtype), None, None, None, None)


def attention_rpe2d_fused(qkv, tkv, tkh, tvv, tvh, scale, max_relative_position=14, dropout_p=0.0, seed=None):
    """qkv (B, N, 3, H, 64) bf16/fp32; tables (2*mr+2, 64) fp32 -> (B, N, H, 64).  dropout_p > 0: attention dropout
    in-kernel (multihead_super.py:145); `seed` names the keep mask (None: drawn from torch's generator state, so
    torch.manual_seed reproduces a run — irpe_fused._new_seed)."""
    if not supported(qkv, dropout_p, max_relative_position, (tkv, tkh, tvv, tvh)):
        raise RuntimeError("cream_amd: fused attention does not cover this shape/dtype/layout "
                           f"(qkv {tuple(qkv.shape)} {qkv.dtype}, strides {qkv.stride()}, dropout {dropout_p})")
    if dropout_p > 0.0 and seed is None:
        from ..irpe_fused import _new_seed
        seed = _new_seed(qkv.device)
    return _FusedAttention.apply(qkv, tkv, tkh, tvv, tvh, scale, max_relative_position, float(dropout_p), int(seed or 0))

"""Attention core of AutoFormer's AttentionSuper with 2-D relative position bias on keys
and values, evaluated through the bucketed identity of SURVEY Appendix B.1 (verified
against multihead_super.py:133-160 to fp32 round-off):

    Lv = q Tkv^T ; Lh = q Tkh^T                                   (N x nb each)
    A[i,j] = s * ( q_i.k_j + Lv[i, iv[i,j]] + Lh[i, ih[i,j]] )      gather   = rpe_index fwd
    P = dropout(softmax_j(A))
    Sv[i,u] = sum_{j: iv[i,j]=u} P[i,j] ; Sh likewise               scatter  = rpe_index bwd
    O_i = sum_j P[i,j] v_j + Sv[i,:] Tvv + Sh[i,:] Tvh

Two executions, both on the HIP kernels of libcream_amd.so:
  * 'bucketed' — dense contractions through the GEMM library, gather/scatter through the
    rpe_index kernels (cream_amd.rpe_index).  Shape-generic (any N, head_dim, dropout).
  * 'fused'    — cream_attn_rpe2d_fwd/bwd: QK^T, the bucket lookups (as a one-hot
    extension of the MFMA contraction), softmax, PV and the bucket sums in one pass,
    nothing of size N^2 touches HBM.  head_dim 64, N = g*g + 1 <= 256, no attention dropout.
'auto' picks 'fused' whenever its constraints hold.
"""
import torch

from .. import rpe_index as _rpe


class _RPEGather(torch.autograd.Function):
    """Y[b,h,i,j] = X[b,h,i,idx[i,j]]; backward = deterministic scatter-add."""

    @staticmethod
    def forward(ctx, x, idx):
        ctx.save_for_backward(idx)
        ctx.nb = x.shape[-1]
        return (_rpe.forward_cpu if x.device.type == "cpu" else _rpe.forward_gpu)(x, idx)

    @staticmethod
    def backward(ctx, gy):
        (idx,) = ctx.saved_tensors
        gy = gy.contiguous()
        B, H, Lq, _ = gy.shape
        if gy.device.type == "cpu":
            gx = gy.new_zeros((B, H, Lq, ctx.nb))
            _rpe.backward_cpu(gx, gy, idx)
        else:
            gx = gy.new_empty((B, H, Lq, ctx.nb))
            _rpe.backward_gpu(gx, gy, idx, accumulate=False)
        return gx, None


class _RPEScatter(torch.autograd.Function):
    """S[b,h,i,u] = sum_{j: idx[i,j]=u} P[b,h,i,j] — the adjoint of _RPEGather (its forward
    is the rpe_index backward kernel and vice versa)."""

    @staticmethod
    def forward(ctx, p, idx, nb):
        ctx.save_for_backward(idx)
        p = p.contiguous()
        B, H, Lq, _ = p.shape
        if p.device.type == "cpu":
            s = p.new_zeros((B, H, Lq, nb))
            _rpe.backward_cpu(s, p, idx)
        else:
            s = p.new_empty((B, H, Lq, nb))
            _rpe.backward_gpu(s, p, idx, accumulate=False)
        return s

    @staticmethod
    def backward(ctx, gs):
        (idx,) = ctx.saved_tensors
        fn = _rpe.forward_cpu if gs.device.type == "cpu" else _rpe.forward_gpu
        return fn(gs, idx), None, None


def rpe_gather(x, idx):
    return _RPEGather.apply(x, idx)


def rpe_scatter(p, idx, nb):
    return _RPEScatter.apply(p, idx, nb)


def _bucketed(qkv, tkv, tkh, tvv, tvh, iv, ih, scale, dropout_p):
    B, N, _, H, D = qkv.shape
    q, k, v = qkv.permute(2, 0, 3, 1, 4).unbind(0)                 # (B, H, N, D) views
    nb = tkv.shape[0]
    attn = q @ k.transpose(-2, -1)
    lv = q @ tkv.t()                                               # (B, H, N, nb)
    lh = q @ tkh.t()
    attn = (attn + rpe_gather(lv, iv) + rpe_gather(lh, ih)) * scale
    p = attn.softmax(dim=-1, dtype=torch.float32)
    if dropout_p > 0.0:
        p = torch.nn.functional.dropout(p, p=dropout_p, training=True)
    out = p.to(v.dtype) @ v
    sv = rpe_scatter(p, iv, nb)
    sh = rpe_scatter(p, ih, nb)
    out = out + (sv.to(tvv.dtype) @ tvv + sh.to(tvh.dtype) @ tvh).to(out.dtype)
    return out.transpose(1, 2)                                     # (B, N, H, D)


def _fused_operands(qkv, tables):
    """The fused kernels read fp32 tables and a bf16/fp32 qkv buffer; under autocast the
    tables are fp32 parameters already."""
    return qkv, tuple(t.float() if t.dtype != torch.float32 else t for t in tables)


def attention_rpe2d(qkv, tkv, tkh, tvv, tvh, iv, ih, scale, dropout_p=0.0, impl='auto',
                    max_relative_position=14):
    """qkv: (B, N, 3, H, D) — q/k/v of head h at [:, :, 0/1/2, h, :]; tables (nb, D);
    iv/ih int32 (N, N) (used by the bucketed execution; the fused kernels regenerate them
    from the grid geometry).  Returns (B, N, H, D)."""
    assert qkv.dim() == 5 and qkv.shape[2] == 3
    if impl not in ('auto', 'fused', 'bucketed'):
        raise ValueError(f"unknown attention impl {impl!r}")
    if impl != 'bucketed' and qkv.is_cuda:
        from . import fused_attention
        qkv_f, tabs = _fused_operands(qkv, (tkv, tkh, tvv, tvh))
        if impl == 'fused' or fused_attention.supported(qkv_f, dropout_p, max_relative_position, tabs):
            return fused_attention.attention_rpe2d_fused(qkv_f, *tabs, scale, max_relative_position, dropout_p=dropout_p)
    elif impl == 'fused':
        raise RuntimeError("cream_amd: the fused attention kernels need device tensors")
    return _bucketed(qkv, tkv, tkh, tvv, tvh, iv, ih, scale, dropout_p)


def attention_plain(qkv, scale, dropout_p=0.0):
    """Softmax attention without relative position terms (relative_position=False)."""
    q, k, v = qkv.permute(2, 0, 3, 1, 4).unbind(0)
    p = ((q @ k.transpose(-2, -1)) * scale).softmax(dim=-1, dtype=torch.float32)
    if dropout_p > 0.0:
        p = torch.nn.functional.dropout(p, p=dropout_p, training=True)
    return (p.to(v.dtype) @ v).transpose(1, 2)

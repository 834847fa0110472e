"""fp32 parity mode of the weight-entangled operators on the framework's OWN kernels (csrc/gemm_f32.hip: exact-fp32
matrix-core products; the fp32 instantiation of the LayerNorm kernels of csrc/block_ops.hip).

`linear` = LinearSuper.forward / qkv_super.forward on the active block of the fp32 SUPER weight, read in place
(AutoFormer/model/module/Linear_super.py:38-54, :71-81; qkv_super.py:45-55, :72-83 — the interleaved q / k / v rows
3 i + j are addressed through the kernel's row map, no regrouped copy); `layer_norm` = LayerNormSuper.forward
(layernorm_super.py:26-37).  Both carry hand-written backwards on the same kernels (dgrad, wgrad + bias column sums,
LayerNorm backward with fixed-order partial sums).  Used by the modules whenever a CUDA fp32 tensor arrives outside
autocast — the configuration the "within 1e-3 of the reference" tests run in.  No CPU / library fallback here: host
tensors keep the module's plain PyTorch formulation (host-logic tests), device tensors always take these kernels.
"""
import ctypes

import torch

from .. import _lib


CALLS = {"linear": 0, "layer_norm": 0, "matmul": 0}       # launches through this module (tests assert which path ran)


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _stream(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def usable(x, *params):
    return (x.is_cuda and x.dtype == torch.float32 and not torch.is_autocast_enabled('cuda')
            and all(p is None or (p.dtype == torch.float32 and p.is_cuda) for p in params))


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, N, K, seg, step):
        lib = _lib.load()
        lead = x.shape[:-1]
        x2 = x.reshape(-1, x.shape[-1])
        if x2.stride(-1) != 1:
            x2 = x2.contiguous()
        M = x2.shape[0]
        assert x2.shape[1] >= K and weight.dim() >= 2 and weight.stride(-1) == 1
        w2 = weight if weight.dim() == 2 else weight.reshape(weight.shape[0], -1)      # conv weight (out, C*ph*pw)
        y = torch.empty((M, N), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(lib.cream_linear_f32_fwd(_p(y), _p(x2), _p(w2), _p(bias), M, N, K, x2.stride(0), w2.stride(0), seg, step,
                                                _stream(x.device)), "cream_linear_f32_fwd")
        ctx.save_for_backward(x2, weight, bias if bias is not None else x2.new_empty(0))
        ctx.dims = (M, N, K, seg, step, tuple(x.shape), bias is not None)
        return y.view(*lead, N)

    @staticmethod
    def backward(ctx, dy):
        x2, weight, bias = ctx.saved_tensors
        M, N, K, seg, step, xshape, has_bias = ctx.dims
        lib = _lib.load()
        dev = dy.device
        dy2 = dy.reshape(M, N).contiguous()
        w2 = weight if weight.dim() == 2 else weight.reshape(weight.shape[0], -1)
        dx = dw = db = None
        with torch.cuda.device(dev):
            st = _stream(dev)
            if ctx.needs_input_grad[0]:
                dx2 = torch.empty((M, K), dtype=torch.float32, device=dev)
                _lib.check(lib.cream_linear_f32_dgrad(_p(dx2), _p(dy2), _p(w2), M, N, K, w2.stride(0), seg, step, st),
                           "cream_linear_f32_dgrad")
                if xshape[-1] != K:                                   # the module sliced x[..., :K]
                    full = torch.zeros((M, xshape[-1]), dtype=torch.float32, device=dev)
                    full[:, :K] = dx2
                    dx2 = full
                dx = dx2.view(xshape)
            if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
                dw2 = torch.zeros_like(w2, memory_format=torch.contiguous_format)          # zeros outside the active slice
                db = torch.zeros_like(bias) if has_bias else None
                _lib.check(lib.cream_linear_f32_wgrad(_p(dw2), _p(db), _p(dy2), _p(x2), M, N, K, x2.stride(0), dw2.stride(0), seg,
                                                      step, st), "cream_linear_f32_wgrad")
                dw = dw2.view(weight.shape)
        return dx, dw, db, None, None, None, None


def linear(x, weight, bias, N, K, seg=0, step=0):
    """y[..., :N] = x[..., :K] . W[wmap(n), :K]^T + bias[:N];  weight / bias are the fp32 SUPER parameters."""
    CALLS["linear"] += 1
    return _Linear.apply(x, weight, bias, int(N), int(K), int(seg), int(step))


class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, E, eps):
        lib = _lib.load()
        x2 = x.reshape(-1, E)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        M = x2.shape[0]
        g, b = weight.detach()[:E].contiguous(), bias.detach()[:E].contiguous()
        y = torch.empty_like(x2)
        mean = torch.empty(M, dtype=torch.float32, device=x.device)
        rstd = torch.empty(M, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(lib.cream_ln_f32_fwd(_p(y), _p(mean), _p(rstd), _p(x2), _p(g), _p(b), M, E, float(eps), _stream(x.device)),
                       "cream_ln_f32_fwd")
        ctx.save_for_backward(x2, mean, rstd, g)
        ctx.dims = (M, E, tuple(x.shape), tuple(weight.shape))
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, mean, rstd, g = ctx.saved_tensors
        M, E, xshape, wshape = ctx.dims
        lib = _lib.load()
        dev = dy.device
        dy2 = dy.reshape(M, E).contiguous()
        dx = torch.empty_like(x2)
        partial = torch.empty((lib.cream_ln_partials(), 3, E), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.cream_ln_f32_bwd(_p(dx), _p(partial), _p(dy2), _p(x2), _p(mean), _p(rstd), _p(g), M, E, _stream(dev)),
                       "cream_ln_f32_bwd")
        sums = partial.sum(dim=0)                                     # fixed-order tree over the 1024 slab partials
        dw = torch.zeros(wshape, dtype=torch.float32, device=dev)
        db = torch.zeros(wshape, dtype=torch.float32, device=dev)
        dw[:E], db[:E] = sums[0], sums[1]
        return dx.view(xshape), dw, db, None, None


def layer_norm(x, weight, bias, E, eps):
    """LayerNorm over the first E channels with the SUPER affine parameters (layernorm_super.py:33-37)."""
    assert x.shape[-1] == E
    CALLS["layer_norm"] += 1
    return _LayerNorm.apply(x, weight, bias, int(E), float(eps))


# ---- batched products of the attention in parity mode (cream_bmm_f32) --------------------------------------------
def _bmm_raw(a, b):
    """a (B0, B1, M, K) . b (B0, B1, K, N) -> fresh contiguous (B0, B1, M, N); any element strides (transposed and
    expanded views are passed as they are)."""
    B0, B1, M, K = a.shape
    N = b.shape[-1]
    assert b.shape == (B0, B1, K, N), (a.shape, b.shape)
    lib = _lib.load()
    c = torch.empty((B0, B1, M, N), dtype=torch.float32, device=a.device)
    if c.numel() == 0:
        return c
    if K == 0:
        return c.zero_()
    I4 = ctypes.c_int64 * 4
    sa = I4(a.stride(2), a.stride(3), a.stride(0), a.stride(1))
    sb = I4(b.stride(2), b.stride(3), b.stride(0), b.stride(1))
    sc = I4(c.stride(2), c.stride(3), c.stride(0), c.stride(1))
    with torch.cuda.device(a.device):
        _lib.check(lib.cream_bmm_f32(_p(c), _p(a), _p(b), M, N, K, sa, sb, sc, B0, B1, _stream(a.device)), "cream_bmm_f32")
    return c


def _as4(t):
    while t.dim() < 4:
        t = t.unsqueeze(0)
    return t


class _Bmm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a4, b4 = _as4(a), _as4(b)
        assert a4.dim() == 4 and b4.dim() == 4, "matmul: at most two batch dimensions"
        B0, B1 = max(a4.shape[0], b4.shape[0]), max(a4.shape[1], b4.shape[1])
        ae, be = a4.expand(B0, B1, *a4.shape[2:]), b4.expand(B0, B1, *b4.shape[2:])
        ctx.save_for_backward(a, b)
        out = _bmm_raw(ae, be)
        nd = max(a.dim(), b.dim())
        return out.reshape(out.shape[4 - nd:]) if nd < 4 else out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        a4, b4, g4 = _as4(a), _as4(b), _as4(g)
        B0, B1 = g4.shape[0], g4.shape[1]
        ae, be = a4.expand(B0, B1, *a4.shape[2:]), b4.expand(B0, B1, *b4.shape[2:])
        da = db = None
        if ctx.needs_input_grad[0]:
            da = _bmm_raw(g4, be.transpose(-1, -2)).sum_to_size(a4.shape).reshape(a.shape)
        if ctx.needs_input_grad[1]:
            db = _bmm_raw(ae.transpose(-1, -2), g4).sum_to_size(b4.shape).reshape(b.shape)
        return da, db


def matmul(a, b):
    """torch.matmul(a, b) for fp32 device operands with up to two (broadcastable) batch dimensions, on cream_bmm_f32:
    q k^T / P v of RPEAttention.forward (rpe_vision_transformer.py:76, :88) and the lookup products of irpe.py:641-644,
    :683-687.  Batch-broadcast operands (a shared-head lookup table) get their gradient as the fixed-order sum of the
    per-item products."""
    CALLS["matmul"] = CALLS.get("matmul", 0) + 1
    return _Bmm.apply(a, b)

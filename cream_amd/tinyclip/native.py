"""TinyCLIP image towers on the framework's own kernels — the run of `ResidualAttentionBlock`s
(TinyCLIP/src/open_clip/model.py:208-315, `Transformer` :342-427) of a ViT tower as ONE autograd node under bf16 autocast:

    x1 = x  + out_proj(attn(ln_1(x)))          attn = plain multi-head attention, heads of 64 (no mask)
    x2 = x1 + c_proj(gelu(c_fc(ln_2(x1))))

on the kernels the AutoFormer block uses (cream_amd.autoformer.block wrappers over the C ABI): LayerNorm passes with the
residual add of the previous branch folded in (csrc/block_ops.hip), the own MFMA GEMMs with bias / erf-GELU / GELU'
epilogues on bf16 operand copies of the weights (csrc/gemm_mfma.hpp), the fused flash-style attention of csrc/irpe_attn.hip
without any relative position term, split-K weight gradients on a side stream whose partials cream_grad_finalize adds
straight into `.grad`.  Round 2 ran the towers' projections, LayerNorms and GELUs through the framework (46 of 85 ms of the
distillation step in a vendor GEMM library, ~20 ms in separate elementwise launches).

fp32 residual stream inside the node (the reference's autocast keeps its stream in the 16-bit dtype; ours is the more
accurate of the two), bf16 GEMM operands, output cast back to the caller's dtype.  The text towers run the same node with
the causal mask applied inside the attention kernels (`causal`: keys j <= i only, model.py:756-762 without an (L, L)
tensor); every fp32 / CPU call and any other mask keeps the composed module path.
"""
import torch

from .. import irpe_fused
from ..autoformer import block as K

_OPS_KEY = '_cream_tc_operands'


class BlockView:
    """What the native block sequence needs to know about one pre-LayerNorm transformer block, whatever its module layout:
    the two LayerNorms, the four (weight, bias) pairs in the order qkv ([q; k; v] rows), output projection, fc1, fc2, the number
    of heads (of 64), and — for DeiT with iRPE (cream_amd.deit_native) — the relative position modules of the attention and its
    dropout rate."""
    __slots__ = ("ln1", "ln2", "pairs", "heads", "rpes", "drop_p", "scale", "drop_path")

    def __init__(self, ln1, ln2, pairs, heads, rpes=(None, None, None), drop_p=0.0, scale=0.125, drop_path=0.0):
        self.ln1, self.ln2, self.pairs, self.heads, self.rpes, self.drop_p, self.scale = ln1, ln2, pairs, heads, rpes, drop_p, scale
        self.drop_path = drop_path        # stochastic depth rate of the block's two branches (rpe_vision_transformer.py:115-116)


def clip_view(blk):
    """TinyCLIP's ResidualAttentionBlock (model.py:208-315): nn.MultiheadAttention packing, c_fc / c_proj."""
    at, mlp = blk.attn, blk.mlp
    return BlockView(blk.ln_1, blk.ln_2, [(at.in_proj_weight, at.in_proj_bias), (at.out_proj.weight, at.out_proj.bias),
                                          (mlp.c_fc.weight, mlp.c_fc.bias), (mlp.c_proj.weight, mlp.c_proj.bias)], at.num_heads)


class BlockOperands:
    """bf16 operand copies of one block: W (out, in) and W^T (in, out) of the qkv projection ([q; k; v] rows), the output
    projection, fc1, fc2, and the biases — written by cream_adamw_step's copy mode in one launch; stale after any optimizer
    step (block._register's global post-step hook) or version change."""

    def __init__(self, view):
        self.params = list(view.pairs)
        dev = self.params[0][0].device
        bf = dict(dtype=torch.bfloat16, device=dev)
        self.w = [torch.empty(tuple(w.shape), **bf) for w, _ in self.params]
        self.wt = [torch.empty((w.shape[1], w.shape[0]), **bf) for w, _ in self.params]
        self.b = [torch.empty(tuple(b.shape), **bf) for _, b in self.params]
        self.key = self._key()
        self.versions = None
        self._table = None
        K._register(self)

    def _key(self):
        return tuple(w.data_ptr() for w, _ in self.params)

    def _versions(self):
        return tuple(p._version for pair in self.params for p in pair)

    def refresh(self):
        if self._table is None:
            jobs = []
            for (w, b), mw, mwt, mb in zip(self.params, self.w, self.wt, self.b):
                jobs.append(K.param_job(w.detach(), mir=mw, mir_t=mwt))
                jobs.append(K.param_job(b.detach(), mir=mb))
            self._table = K.JobTable(jobs, self.w[0].device)
        self._table.launch(update=False)
        self.versions = self._versions()

    def stale(self):
        return self.versions != self._versions()


def operands(blk, view=None):
    ops = blk.__dict__.get(_OPS_KEY)
    if ops is None or ops.key != ops._key():
        ops = blk.__dict__[_OPS_KEY] = BlockOperands(view if view is not None else clip_view(blk))
    if ops.stale():
        ops.refresh()
    return ops


def supported(transformer, x, attn_mask, causal=False):
    """`causal`: the caller vouches that `attn_mask` is the additive upper-triangular -inf mask (TextEncoder's buffer)."""
    if not len(transformer.resblocks):
        return False
    if not ((attn_mask is None or causal) and x.is_cuda and x.dim() == 3
            and torch.is_autocast_enabled('cuda') and torch.get_autocast_dtype('cuda') == torch.bfloat16
            and transformer.head_dim == 64 and transformer.width % 8 == 0 and x.shape[1] <= 2048):
        return False
    if not all(_block_supported(blk) for blk in transformer.resblocks):
        return False
    if torch.is_grad_enabled():
        # the native backward writes a `.grad` for every parameter of every block: a tower with frozen parameters that is
        # differentiated through (a locked tower, partial fine-tuning) keeps the composed path
        flags = [p.requires_grad for p in transformer.resblocks.parameters()]
        if (x.requires_grad or any(flags)) and not all(flags):
            return False
    return True


def _block_supported(blk):
    """Every block is looked at, not the first alone."""
    return (isinstance(blk.mlp.gelu, torch.nn.GELU) and getattr(blk.mlp.gelu, 'approximate', 'none') == 'none'
            and isinstance(blk.ln_attn, torch.nn.Identity) and blk.attn.in_proj_weight.dtype == torch.float32
            and blk.mlp.c_fc.weight.shape[0] % 8 == 0)


def _attn_meta(view, L, dev, training):
    """(terms, dropout rate, seed) of a block's attention: the operands of the rpe terms (irpe_fused._term) and one keep-mask seed
    per forward call; all None / 0 for plain attention."""
    if all(r is None for r in view.rpes) and not (training and view.drop_p):
        return irpe_fused._NO_TERMS, 0.0, 0
    terms = tuple(irpe_fused._term(r, L, dev) for r in view.rpes)
    p = float(view.drop_p) if training else 0.0
    return terms, p, (irpe_fused._new_seed(dev) if p else 0)


def _path_scales(view, B, dev, training):
    """Stochastic depth (timm's DropPath as the reference's blocks use it: x + drop_path(branch(x))): per-sample factors
    mask_b / keep of the attention branch and of the MLP branch, drawn from the device generator in the module's order; None when
    inactive.  The LayerNorm / residual kernels take them as `sample_scale` (the AutoFormer block's mechanism)."""
    if not (training and view.drop_path):
        return None, None
    keep = 1.0 - float(view.drop_path)
    draw = lambda: torch.floor(keep + torch.rand(B, device=dev, dtype=torch.float32)) / keep     # noqa: E731
    return draw(), draw()


def _block_forward(view, ops, x, pend, pend_scale, B, L, keep, causal, meta, s_attn):
    """x (M, D) fp32 stream (or, with pend = previous branch output f and its per-sample factors, the stream before that add).
    -> (x1, f, saved)"""
    M, D = x.shape
    H = view.heads
    F_ = view.pairs[2][0].shape[0]
    (wqkv, wo, w1, w2), (bqkv, bo, b1, b2) = ops.w, ops.b
    ln1, ln2 = view.ln1, view.ln2
    terms, drop_p, seed = meta
    if pend is None:
        xin = x
        a, mean1, rstd1 = K.ln_fwd(x, ln1.weight, ln1.bias, ln1.eps)
    else:
        xin, a, mean1, rstd1 = K.add_ln_fwd(x, pend, pend_scale, L, ln1.weight, ln1.bias, ln1.eps)
    qkv = K.linear_fwd(a, wqkv, bqkv, 3 * D, D)
    o, lse, sv = irpe_fused.fwd_core(qkv.view(B, L, 3, H, 64), view.scale, terms, drop_p, seed, causal)
    p = K.linear_fwd(o.view(M, D), wo, bo, D, D)
    x1, c, mean2, rstd2 = K.add_ln_fwd(xin, p, s_attn, L, ln2.weight, ln2.bias, ln2.eps)
    gp, g = K.linear_gelu_fwd(c, w1, b1, F_, D, want_grad=keep)      # (the frozen teacher writes no gelu')
    f = K.linear_fwd(g, w2, b2, D, F_)
    saved = (xin, mean1, rstd1, a, qkv, o, lse, x1, mean2, rstd2, c, gp, g, sv if sv is not None else lse) if keep else None
    return x1, f, saved


def _add_grad(param, g):
    param.grad = g if param.grad is None else param.grad + g


def _block_backward(blk, view, ops, saved, dx2, df, pb2, B, L, want_prev, causal, meta, s_attn, s_prev):
    """dx2 (M, D) fp32 gradient of the block's output stream, df (M, D) bf16 = gradient of the c_proj output with its
    per-slab column sums pb2 = (tensor, nparts, pstride, offset).  -> (dx, df_prev, pb2_prev)"""
    xin, mean1, rstd1, a, qkv, o, lse, x1, mean2, rstd2, c, gp, g, sv = saved
    M, D = xin.shape
    H = view.heads
    (Wqkv, Bqkv), (Wo, Bo), (W1, B1), (W2, B2) = view.pairs
    F_ = W1.shape[0]
    wqkv_t, wo_t, w1_t, w2_t = ops.wt
    ln1, ln2 = view.ln1, view.ln2
    terms, drop_p, seed = meta
    jobs = K.GradJobs()
    pw2, _ = K.wgrad_parts_async(df, g)
    jobs.add(W2, pw2, pw2.shape[0], D * F_, D, F_)
    jobs.add(B2, pb2[0], pb2[1], pb2[2], 1, D, src_offset=pb2[3])
    dh, pb1 = K.linear_dgrad_mul(df, w2_t, gp, D, F_)
    pw1, _ = K.wgrad_parts_async(dh, c)
    jobs.add(W1, pw1, pw1.shape[0], F_ * D, F_, D)
    jobs.add(B1, pb1, pb1.shape[0], F_, 1, F_)
    dc = K.linear_dgrad(dh, w1_t, F_, D)
    dx1, dp, pl2 = K.ln_bwd_raw(dc, x1, mean2, rstd2, ln2.weight, dx2, s_attn, L, True)
    P = pl2.shape[0]
    jobs.add(ln2.weight, pl2, P, 3 * D, 1, D)
    jobs.add(ln2.bias, pl2, P, 3 * D, 1, D, src_offset=D)
    jobs.add(Bo, pl2, P, 3 * D, 1, D, src_offset=2 * D)
    pwp, _ = K.wgrad_parts_async(dp, o.view(M, D))
    jobs.add(Wo, pwp, pwp.shape[0], D * D, D, D)
    do = K.linear_dgrad(dp, wo_t, D, D)
    dqkv, table_grads = irpe_fused.bwd_core(do.view(B, L, D), qkv.view(B, L, 3, H, 64), o, lse, sv, view.scale, terms, drop_p, seed, causal)
    for rpe, tg in zip(view.rpes, table_grads):           # lookup tables of the rpe terms: accumulated like every other gradient of the node
        if tg is not None:
            _add_grad(rpe.lookup_table_bias if rpe.mode == "bias" else rpe.lookup_table_weight, tg)
    dqkv2d = dqkv.view(M, 3 * D)
    pwq, pbq = K.wgrad_parts_async(dqkv2d, a, want_bias=True)
    jobs.add(Wqkv, pwq, pwq.shape[0], 3 * D * D, 3 * D, D)
    jobs.add(Bqkv, pbq, pbq.shape[0], 3 * D, 1, 3 * D)
    da = K.linear_dgrad(dqkv2d, wqkv_t, 3 * D, D)
    dx, df_prev, pl1 = K.ln_bwd_raw(da, xin, mean1, rstd1, ln1.weight, dx1, s_prev, L, want_prev)
    jobs.add(ln1.weight, pl1, P, 3 * D, 1, D)
    jobs.add(ln1.bias, pl1, P, 3 * D, 1, D, src_offset=D)
    K.finalize_on_side_stream(jobs, blk, [df, g, pw2, pb2[0], dh, c, pw1, pb1, pl2, dp, o, pwp, dqkv, a, pwq, pbq, pl1])
    return dx, df_prev, (pl1, P, 3 * D, 2 * D)


class TowerStack(torch.autograd.Function):
    """x (B, L, D) any float dtype -> (B, L, D) same dtype: all blocks of a tower."""

    @staticmethod
    def forward(ctx, x, blks, view_of, causal, *params):
        B, L, D = x.shape
        M = B * L
        blks = list(blks)
        views = [view_of(blk) for blk in blks]
        keep = any(ctx.needs_input_grad)            # (False under no_grad: the frozen teacher saves nothing)
        cur = x.reshape(M, D).float().contiguous()
        pend = pend_scale = None
        saved, metas, scales = [], [], []
        for blk, view in zip(blks, views):
            meta = _attn_meta(view, L, x.device, blk.training)
            s_attn, s_mlp = _path_scales(view, B, x.device, blk.training)
            x1, f, sv = _block_forward(view, operands(blk, view), cur, pend, pend_scale, B, L, keep, causal, meta, s_attn)
            if keep:
                saved.extend(sv)
            metas.append(meta)
            scales.append((s_attn, s_mlp))
            cur, pend, pend_scale = x1, f, s_mlp
        out = K.residual_add(cur, pend, pend_scale, L * D)
        ctx.blks, ctx.views, ctx.metas, ctx.dims, ctx.nsaved = blks, views, metas, (B, L, D), (len(saved) // len(blks) if keep else 0)
        ctx.scales = scales
        ctx.in_dtype, ctx.causal = x.dtype, causal
        if keep:
            ctx.save_for_backward(*saved)
        return out.view(B, L, D).to(x.dtype)

    @staticmethod
    def backward(ctx, dout):
        blks, (B, L, D), ns = ctx.blks, ctx.dims, ctx.nsaved
        tens = ctx.saved_tensors
        M = B * L
        dx = dout.reshape(M, D).float().contiguous()
        df, part = K.scale_cast_colsum(dx, ctx.scales[-1][1], L)
        pb2 = (part, part.shape[0], D, 0)
        for i in range(len(blks) - 1, -1, -1):
            blk = blks[i]
            view = ctx.views[i]
            dx, df, pb2 = _block_backward(blk, view, operands(blk, view), tens[i * ns:(i + 1) * ns], dx, df, pb2, B, L, i > 0, ctx.causal,
                                          ctx.metas[i], ctx.scales[i][0], ctx.scales[i - 1][1] if i > 0 else None)
        K.join_side_stream(dx.device)
        return (dx.view(B, L, D).to(ctx.in_dtype), None, None, None) + (None,) * (len(ctx.needs_input_grad) - 4)


def tower(transformer, x, causal=False):
    """Run `transformer.resblocks` natively.  The parameters are passed to the node only so that autograd schedules its
    backward (they receive their gradients in place, announced through block.on_grads_ready)."""
    params = [p for p in transformer.parameters()]
    return TowerStack.apply(x, transformer.resblocks, clip_view, bool(causal), *params)


def stack(blks, view_of, x, causal=False):
    """The same node for any run of pre-LayerNorm blocks described by `view_of(blk) -> BlockView` (cream_amd.deit_native)."""
    params = [p for blk in blks for p in blk.parameters()]
    return TowerStack.apply(x, blks, view_of, bool(causal), *params)

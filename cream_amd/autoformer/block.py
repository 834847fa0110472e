"""The run of supernet transformer blocks as a single autograd node on the native path — the bf16
throughput execution of TransformerEncoderLayer.forward
(AutoFormer/model/supernet_transformer.py:251-287), per block:

    x1 = x  + drop_path(proj(attn(LN1(x))))
    x2 = x1 + drop_path(fc2(gelu(fc1(LN2(x1)))))

What runs where:
  * LayerNorm (fwd/bwd), GELU (fwd/bwd), residual add + drop-path scale, bias-gradient column
    sums: the fused HBM passes of csrc/block_ops.hip (fp32 residual stream, bf16 GEMM operands,
    every activation crosses HBM once per direction);
  * attention core: csrc/attn_rpe2d.hip (nothing of size N^2 in HBM);
  * the dense projections: the GEMM library (hipBLASLt, dispatched from the C ABI with the offline
    kernel table: csrc/gemm_lt.cpp) on bf16 MIRRORS of the fp32 master weights, read in place as
    `W[:out, :in]` (leading dimension = super width); weight gradients are split-K batched GEMMs
    (K = 25k tokens split 8 ways) whose partial products `cream_grad_finalize` adds in fp32 into
    the active slice of the fp32 `.grad`.
Two drivers over the SAME kernels: `NATIVE_BLOCK` (default) = one call into the C ABI per block and
direction (csrc/block_seq.cpp; weight gradients + finalisation on a side stream), or op by op from
here (`_block_forward` / `_block_backward`; used by the kernel-timing pass of bench.py and by the
tests that pin the native sequencing bit for bit).  The backward is written by hand; parameter
gradients are accumulated into `p.grad` directly and announced through the `on_grads_ready` hooks
(the gradient reducer starts a block's all-reduce the moment its last gradient exists).
"""
import ctypes

import torch

from .. import _lib, timing
from . import fused_attention

_WGRAD_SPLIT = 8


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# ---- thin wrappers over the C ABI ------------------------------------------------------------
def ln_fwd(x2d, gamma, beta, eps):
    M, E = x2d.shape
    y = torch.empty((M, E), dtype=torch.bfloat16, device=x2d.device)
    mean = torch.empty(M, dtype=torch.float32, device=x2d.device)
    rstd = torch.empty(M, dtype=torch.float32, device=x2d.device)
    with timing.region("ln_fwd", nbytes=M * E * 6):
        _lib.check(_lib.load().cream_ln_fwd(_p(y), _p(mean), _p(rstd), _p(x2d), _p(gamma), _p(beta), M, E,
                                           float(eps), _stream()), "cream_ln_fwd")
    return y, mean, rstd


def add_ln_fwd(x2d, res, sample_scale, rows_per_sample, gamma, beta, eps):
    """x1 = x + s_b * res (fp32), y = LN(x1) (bf16): the residual add and the next LayerNorm in
    one pass.  -> (x1, y, mean, rstd)"""
    M, E = x2d.shape
    x1 = torch.empty_like(x2d)
    y = torch.empty((M, E), dtype=torch.bfloat16, device=x2d.device)
    mean = torch.empty(M, dtype=torch.float32, device=x2d.device)
    rstd = torch.empty(M, dtype=torch.float32, device=x2d.device)
    with timing.region("add_ln_fwd", nbytes=M * E * (4 + 2 + 4 + 2)):
        _lib.check(_lib.load().cream_add_ln_fwd(_p(x1), _p(y), _p(mean), _p(rstd), _p(x2d), _p(res), _p(sample_scale),
                                               rows_per_sample, _p(gamma), _p(beta), M, E, float(eps), _stream()),
                   "cream_add_ln_fwd")
    return x1, y, mean, rstd


def ln_bwd_raw(dy, x2d, mean, rstd, gamma, dres, sample_scale, rows_per_sample, want_scaled):
    """-> (dx, dx_scaled or None, partial (P, 3, E): per-slab [dgamma, dbeta, colsum(dx_scaled)])"""
    M, E = x2d.shape
    lib = _lib.load()
    dx = torch.empty((M, E), dtype=torch.float32, device=x2d.device)
    dxs = torch.empty((M, E), dtype=torch.bfloat16, device=x2d.device) if want_scaled else None
    partial = torch.empty((lib.cream_ln_partials(), 3, E), dtype=torch.float32, device=x2d.device)
    with timing.region("ln_bwd", nbytes=M * E * (2 + 4 + 4 + 4 + (2 if want_scaled else 0))):
        _lib.check(lib.cream_ln_bwd(_p(dx), _p(dxs), _p(partial), _p(dy), _p(x2d), _p(mean), _p(rstd), _p(gamma),
                                    _p(dres), _p(sample_scale), rows_per_sample, M, E, _stream()), "cream_ln_bwd")
    return dx, dxs, partial


def ln_bwd(dy, x2d, mean, rstd, gamma, dres, sample_scale, rows_per_sample, want_scaled):
    """-> (dx, dx_scaled, (3, E) sums [dgamma, dbeta, colsum(dx_scaled)])"""
    dx, dxs, partial = ln_bwd_raw(dy, x2d, mean, rstd, gamma, dres, sample_scale, rows_per_sample, want_scaled)
    return dx, dxs, partial.sum(dim=0)


def gelu_fwd(h):
    g = torch.empty_like(h)
    with timing.region("gelu_fwd", nbytes=h.numel() * 4):
        _lib.check(_lib.load().cream_gelu_fwd(_p(g), _p(h), h.numel(), _stream()), "cream_gelu_fwd")
    return g


def gelu_bwd(dg, h):
    dh = torch.empty_like(h)
    with timing.region("gelu_bwd", nbytes=h.numel() * 6):
        _lib.check(_lib.load().cream_gelu_bwd(_p(dh), _p(dg), _p(h), h.numel(), _stream()), "cream_gelu_bwd")
    return dh


def residual_add(x2d, y, sample_scale, per_sample):
    out = torch.empty_like(x2d)
    with timing.region("residual_add", nbytes=x2d.numel() * 10):
        _lib.check(_lib.load().cream_residual_add(_p(out), _p(x2d), _p(y), _p(sample_scale), x2d.numel(), per_sample,
                                                  _stream()), "cream_residual_add")
    return out


def scale_cast(x2d, sample_scale, per_sample):
    out = torch.empty(x2d.shape, dtype=torch.bfloat16, device=x2d.device)
    with timing.region("scale_cast", nbytes=x2d.numel() * 6):
        _lib.check(_lib.load().cream_scale_cast(_p(out), _p(x2d), _p(sample_scale), x2d.numel(), per_sample,
                                                _stream()), "cream_scale_cast")
    return out


def colsum(a):
    M, C = a.shape
    lib = _lib.load()
    partial = torch.empty((lib.cream_colsum_slabs(M), C), dtype=torch.float32, device=a.device)
    with timing.region("colsum", nbytes=a.numel() * 2):
        _lib.check(lib.cream_colsum(_p(partial), _p(a), M, C, _stream()), "cream_colsum")
    return partial.sum(dim=0)


def gelu_bwd_colsum(dg, h):
    """-> (dh, partial (slabs, F)): dh = dg * gelu'(h) and its per-slab column sums (fc1 bias)."""
    M, C = h.shape
    lib = _lib.load()
    dh = torch.empty_like(h)
    partial = torch.empty((lib.cream_colsum128_slabs(M), C), dtype=torch.float32, device=h.device)
    with timing.region("gelu_bwd", nbytes=h.numel() * 6):
        _lib.check(lib.cream_gelu_bwd_colsum(_p(dh), _p(partial), _p(dg), _p(h), M, C, _stream()),
                   "cream_gelu_bwd_colsum")
    return dh, partial


def scale_cast_colsum(x2d, sample_scale, rows_per_sample):
    """-> (bf16(s_b * x), partial (slabs, C)) — the branch-output gradient and its column sums."""
    M, C = x2d.shape
    lib = _lib.load()
    out = torch.empty((M, C), dtype=torch.bfloat16, device=x2d.device)
    partial = torch.empty((lib.cream_colsum128_slabs(M), C), dtype=torch.float32, device=x2d.device)
    with timing.region("scale_cast", nbytes=x2d.numel() * 6):
        _lib.check(lib.cream_scale_cast_colsum(_p(out), _p(partial), _p(x2d), _p(sample_scale), rows_per_sample,
                                               M, C, _stream()), "cream_scale_cast_colsum")
    return out, partial


def colsum128(a):
    """per-slab column sums (slabs, C) fp32 of a bf16 (M, C) matrix"""
    M, C = a.shape
    lib = _lib.load()
    partial = torch.empty((lib.cream_colsum128_slabs(M), C), dtype=torch.float32, device=a.device)
    with timing.region("colsum", nbytes=a.numel() * 2):
        _lib.check(lib.cream_colsum128(_p(partial), _p(a), M, C, _stream()), "cream_colsum128")
    return partial


def wgrad_parts(dy, x):
    """Split-K weight gradient: (s, out, in) partial products dy_s^T x_s over s slices of the
    token dimension (the library's single-pass TN GEMM leaves most CUs idle on a 25k-deep
    contraction); the slices are added by cream_grad_finalize."""
    M = dy.shape[0]
    s = _WGRAD_SPLIT
    while M % s:
        s //= 2
    return torch.bmm(dy.view(s, M // s, -1).transpose(1, 2), x.view(s, M // s, -1))


# ---- dense projections: native dispatch (csrc/gemm_lt.cpp) ----------------------------------------
# The GEMM library is called from the C ABI with plans cached per problem signature and the
# offline kernel selection of cream_amd/tuning/*.csv; through the framework each GEMM cost ~27 us
# of host time (tunable-op lookup), 24 GEMMs per block: the step was launch-bound.
NATIVE_GEMM = True
_GEMM_WS_BYTES = 128 << 20
_gemm_ws = {}              # (device, stream handle) -> workspace tensor (kept alive)
_gemm_tables = set()


def gemm_table_load(path):
    """Register a kernel-selection table with the native dispatcher (idempotent)."""
    import os
    path = os.path.abspath(path)
    if path in _gemm_tables:
        return 0
    n = _lib.load().cream_gemm_table_load(path.encode())
    if n < 0:
        raise RuntimeError(f"cream_amd: cannot read GEMM table {path}")
    _gemm_tables.add(path)
    return n


def _gemm_stream(device):
    """Current stream handle, with a GEMM workspace registered for it."""
    st = torch.cuda.current_stream(device)
    h = st.cuda_stream
    key = (device.index, h)
    if key not in _gemm_ws:
        ws = torch.empty(_GEMM_WS_BYTES, dtype=torch.uint8, device=device)
        _lib.check(_lib.load().cream_gemm_set_workspace(ctypes.c_void_p(h), _p(ws), ws.numel()), "cream_gemm_set_workspace")
        _gemm_ws[key] = ws
    return ctypes.c_void_p(h)


def linear_fwd(x, w, bias, N, K, out=None):
    """out (M, N) = x (M, K) . W[:N, :K]^T + bias[:N]; w: bf16 (rows, ld) matrix read in place."""
    M = x.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=torch.bfloat16, device=x.device)
    if not NATIVE_GEMM:
        return torch.addmm(bias[:N], x, w[:N, :K].t(), out=out)
    _lib.check(_lib.load().cream_linear_fwd(_p(out), _p(x), _p(w), _p(bias), M, N, K, w.stride(0), _gemm_stream(x.device)),
               "cream_linear_fwd")
    return out


def linear_dgrad(dy, w, N, K, out=None):
    """dx (M, K) = dy (M, N) . W[:N, :K]"""
    M = dy.shape[0]
    if out is None:
        out = torch.empty((M, K), dtype=torch.bfloat16, device=dy.device)
    if not NATIVE_GEMM:
        return torch.mm(dy, w[:N, :K], out=out)
    _lib.check(_lib.load().cream_linear_dgrad(_p(out), _p(dy), _p(w), M, N, K, w.stride(0), _gemm_stream(dy.device)),
               "cream_linear_dgrad")
    return out


def _wgrad_split(M):
    s = _WGRAD_SPLIT
    while M % s:
        s //= 2
    return s


def linear_wgrad_parts(dy, x, out=None):
    """(S, N, K) partial products dy_s^T x_s over S slices of the token dimension."""
    M, N = dy.shape
    K = x.shape[1]
    s = _wgrad_split(M)
    if out is None:
        out = torch.empty((s, N, K), dtype=torch.bfloat16, device=dy.device)
    if not NATIVE_GEMM:
        return torch.bmm(dy.view(s, M // s, -1).transpose(1, 2), x.view(s, M // s, -1), out=out)
    _lib.check(_lib.load().cream_linear_wgrad_parts(_p(out), _p(dy), _p(x), M, N, K, s, _gemm_stream(dy.device)),
               "cream_linear_wgrad_parts")
    return out


# The weight-gradient GEMMs are off the critical path of a block's backward (only the gradient
# finalisation at its end reads them) and MFMA-bound, while the passes between the dgrad GEMMs
# (GELU', LayerNorm', column sums) are HBM-bound: enqueued on a second HIP stream they fill the
# matrix cores while the main stream streams memory, and the tails of the small-N GEMMs overlap.
_side_streams = {}
WGRAD_SIDE_STREAM = True


def _side_stream(device):
    st = _side_streams.get(device)
    if st is None:
        st = _side_streams[device] = torch.cuda.Stream(device=device)
    return st


def wgrad_parts_async(dy, x):
    """wgrad_parts on the side stream: the output is allocated on the caller's stream (so the
    caching allocator never hands it out while the side stream still writes it: the caller joins
    the side stream before the partials are consumed), the operands are complete at this point of
    the caller's stream (event)."""
    if not WGRAD_SIDE_STREAM:
        return linear_wgrad_parts(dy, x)
    out = torch.empty((_wgrad_split(dy.shape[0]), dy.shape[1], x.shape[1]), dtype=dy.dtype, device=dy.device)
    main = torch.cuda.current_stream(dy.device)
    side = _side_stream(dy.device)
    side.wait_event(main.record_event())
    with torch.cuda.stream(side):
        linear_wgrad_parts(dy, x, out=out)
    return out


def join_side_stream(device):
    if WGRAD_SIDE_STREAM:
        torch.cuda.current_stream(device).wait_stream(_side_stream(device))


def finalize_on_side_stream(jobs, blk, tensors):
    """Gradient finalisation of a block (and the announcement of its gradients to the reducer) on
    the side stream, behind the block's weight-gradient GEMMs: the main stream goes straight on to
    the previous block.  `tensors`: everything the side stream reads — they were allocated on the
    main stream, so the allocator is told not to recycle them before the side stream is done."""
    dev = tensors[0].device
    if not WGRAD_SIDE_STREAM:
        jobs.launch()
        _notify(blk)
        return
    main = torch.cuda.current_stream(dev)
    side = _side_stream(dev)
    side.wait_event(main.record_event())
    with torch.cuda.stream(side):
        jobs.launch()
        _notify(blk)
    for t in tensors:
        t.record_stream(side)


def _notify(blk):
    if _grad_ready_hooks:
        params = [p for p in blk.parameters() if p.requires_grad]
        for fn in _grad_ready_hooks:
            fn(params)


class GradJobs:
    """The gradient finalisation of one block: every `add` names an fp32 gradient (the active
    slice of a super-weight's .grad) and the partial sums that go into it; `launch` adds them all
    in ONE kernel (cream_grad_finalize), fixed summation order."""

    def __init__(self):
        self.jobs = (_lib.GradJob * _lib.MAX_GRAD_JOBS)()
        self.n = 0
        self.keep = []

    def add(self, param, src, nparts, pstride, rows, cols, interleave=0, src_offset=0):
        if param.grad is None:
            param.grad = torch.zeros_like(param, memory_format=torch.contiguous_format)
        g = param.grad
        j = self.jobs[self.n]
        j.dst = g.data_ptr()
        j.src = src.data_ptr() + src_offset * src.element_size()
        j.ld = g.stride(0) if g.dim() == 2 else cols
        j.pstride = pstride
        j.nparts, j.rows, j.cols = nparts, rows, cols
        j.interleave = interleave
        j.src_bf16 = 1 if src.dtype == torch.bfloat16 else 0
        self.n += 1
        self.keep.append(src)

    def launch(self):
        with timing.region("grad_finalize"):
            _lib.check(_lib.load().cream_grad_finalize(ctypes.cast(self.jobs, ctypes.c_void_p), self.n, _stream()),
                       "cream_grad_finalize")
        self.keep.clear()


def wgrad(dy, x):
    """dW (out, in) fp32 = dy^T x with the token dimension split _WGRAD_SPLIT ways (the
    library's single-pass TN GEMM leaves most CUs idle on a 25k-deep contraction)."""
    M = dy.shape[0]
    s = _WGRAD_SPLIT
    while M % s:
        s //= 2
    part = torch.bmm(dy.view(s, M // s, -1).transpose(1, 2), x.view(s, M // s, -1))
    return part.sum(dim=0, dtype=torch.float32)


# ---- bf16 mirrors of the fp32 master weights --------------------------------------------------
class Mirror:
    """bf16 copies of parameters, refreshed when the parameter's version counter moves (the
    optimizer step bumps it).  `refresh_all` converts everything in a few fused launches."""

    def __init__(self):
        self._m = {}

    def get(self, p):
        e = self._m.get(id(p))
        if e is None or e[1] != p._version or e[0].device != p.device:
            m = p.detach().to(torch.bfloat16) if e is None or e[0].device != p.device else e[0].copy_(p.detach())
            self._m[id(p)] = e = (m, p._version, p)
        return e[0]

    def refresh_all(self, force=False):
        """Re-convert the mirrored parameters.  force=True ignores the version counters: fused
        optimizer kernels (torch._fused_adamw_) update parameters without bumping them, so the
        optimizer-step hook installed by engine.build_optimizer always forces."""
        stale = [e for e in self._m.values() if force or e[1] != e[2]._version]
        if stale:
            torch._foreach_copy_([e[0] for e in stale], [e[2].detach() for e in stale])
            for e in stale:
                self._m[id(e[2])] = (e[0], e[2]._version, e[2])


MIRROR = Mirror()
_grad_ready_hooks = []


def on_grads_ready(fn):
    """Register fn(params) to be called when a block has finished writing these `.grad`s."""
    _grad_ready_hooks.append(fn)
    return fn


def remove_grads_ready(fn):
    if fn in _grad_ready_hooks:
        _grad_ready_hooks.remove(fn)


def _acc(p, sl, g):
    """p.grad[sl] += g (fp32), creating a zero gradient if the parameter has none yet."""
    if p.grad is None:
        p.grad = torch.zeros_like(p, memory_format=torch.contiguous_format)
    p.grad[sl].add_(g)


def supported(blk, x):
    a = blk.attn
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 3 and blk.normalize_before and not blk.scale
            and a.relative_position and a.change_qkv and not a.fc_scale
            and (not blk.training or (a.attn_drop.p == 0.0 and (blk.sample_dropout or 0.0) == 0.0
                                      and (blk.sample_attn_dropout or 0.0) == 0.0 and a.proj_drop.p == 0.0))
            and fused_attention.grid_of(x.shape[1], a.max_relative_position) is not None
            and a.rel_pos_embed_k.embeddings_table_v.shape[1] == 64
            and blk.sample_embed_dim % 8 == 0 and blk.sample_ffn_embed_dim_this_layer % 8 == 0
            and blk.sample_out_dim == blk.sample_embed_dim)


def _tables(at):
    return (at.rel_pos_embed_k.embeddings_table_v, at.rel_pos_embed_k.embeddings_table_h,
            at.rel_pos_embed_v.embeddings_table_v, at.rel_pos_embed_v.embeddings_table_h)


def _block_forward(blk, x2d, pend, dp1, B, N):
    """One block up to (not including) its last residual add.
    x2d   (M, E) fp32 residual stream entering the block — or, with pend = (f_prev, s_prev), the
          stream BEFORE the previous block's last residual add: that add rides on this block's
          first LayerNorm pass (x = x2d + s_prev * f_prev).
    -> (x1, f, saved): the block's output is x1 + s2 * f, left pending for the next consumer."""
    M, E = x2d.shape
    at = blk.attn
    H = at.sample_num_heads
    Q = at.sample_qk_embed_dim
    F_ = blk.sample_ffn_embed_dim_this_layer
    mr = at.max_relative_position
    mir = MIRROR.get
    ln1, ln2 = blk.attn_layer_norm, blk.ffn_layer_norm

    if pend is None:
        x = x2d
        a, mean1, rstd1 = ln_fwd(x, ln1.weight[:E], ln1.bias[:E], ln1.eps)
    else:
        x, a, mean1, rstd1 = add_ln_fwd(x2d, pend[0], pend[1], N, ln1.weight[:E], ln1.bias[:E], ln1.eps)
    # qkv rows regrouped [q | k | v] from the interleaved super weight (qkv_super.py:72-77);
    # bias is the plain prefix (qkv_super.py:80-83)
    wqkv = mir(at.qkv.weight)[:3 * Q, :E].view(Q, 3, E).transpose(0, 1).reshape(3 * Q, E)
    qkv = linear_fwd(a, wqkv, mir(at.qkv.bias), 3 * Q, E)
    tabs = tuple(t.detach() for t in _tables(at))
    o, lse, sp = fused_attention.attn_fwd_raw(qkv.view(B, N, 3, H, 64), *tabs, at.sample_scale, mr)
    p = linear_fwd(o.view(M, Q), mir(at.proj.weight), mir(at.proj.bias), E, Q)
    x1, c, mean2, rstd2 = add_ln_fwd(x, p, dp1, N, ln2.weight[:E], ln2.bias[:E], ln2.eps)
    h = linear_fwd(c, mir(blk.fc1.weight), mir(blk.fc1.bias), F_, E)
    g = gelu_fwd(h)
    f = linear_fwd(g, mir(blk.fc2.weight), mir(blk.fc2.bias), E, F_)
    dims = (B, N, E, H, Q, F_, mr, float(at.sample_scale))
    return x1, f, dims, (x, mean1, rstd1, a, wqkv, qkv, o, lse, sp, x1, mean2, rstd2, c, h, g)


def _block_backward(blk, dims, dp1, saved, dx2, df, pb2, prev_scale, want_prev):
    """Backward of one block.  dx2 (M, E) fp32: gradient of the block's output stream;
    df (M, E) bf16 = s2 * dx2: gradient of the fc2 output, pb2 its per-slab column sums as
    (tensor, nparts, pstride, offset) — both produced by whoever consumed this block's output.
    want_prev: the block's input was itself a pending sum x_prev1 + s_prev * f_prev (the previous
    block of the stack): then the last pass also emits df_prev = bf16(prev_scale * dx) and its
    column sums, i.e. the previous block's (df, pb2).   -> (dx, df_prev, pb2_prev)"""
    at = blk.attn
    B, N, E, H, Q, F_, mr, scale = dims
    M = B * N
    x, mean1, rstd1, a, wqkv, qkv, o, lse, sp, x1, mean2, rstd2, c, h, g = saved
    mir = MIRROR.get
    ln1, ln2 = blk.attn_layer_norm, blk.ffn_layer_norm

    jobs = GradJobs()
    # ---- MLP branch -----------------------------------------------------------------------
    pw2 = wgrad_parts_async(df, g)
    jobs.add(blk.fc2.weight, pw2, pw2.shape[0], E * F_, E, F_)
    jobs.add(blk.fc2.bias, pb2[0], pb2[1], pb2[2], 1, E, src_offset=pb2[3])
    dg = linear_dgrad(df, mir(blk.fc2.weight), E, F_)
    dh, pb1 = gelu_bwd_colsum(dg, h)
    pw1 = wgrad_parts_async(dh, c)
    jobs.add(blk.fc1.weight, pw1, pw1.shape[0], F_ * E, F_, E)
    jobs.add(blk.fc1.bias, pb1, pb1.shape[0], F_, 1, F_)
    dc = linear_dgrad(dh, mir(blk.fc1.weight), F_, E)
    # dx1 = dx2 + dLN2(dc); dp = s1 * dx1 is the gradient of the proj output, and its column sums
    # (proj bias) come out of the same pass
    dx1, dp, pl2 = ln_bwd_raw(dc, x1, mean2, rstd2, ln2.weight[:E], dx2, dp1, N, True)
    P = pl2.shape[0]
    jobs.add(ln2.weight, pl2, P, 3 * E, 1, E)
    jobs.add(ln2.bias, pl2, P, 3 * E, 1, E, src_offset=E)
    jobs.add(at.proj.bias, pl2, P, 3 * E, 1, E, src_offset=2 * E)

    # ---- attention branch ---------------------------------------------------------------------
    pwp = wgrad_parts_async(dp, o.view(M, Q))
    jobs.add(at.proj.weight, pwp, pwp.shape[0], E * Q, E, Q)
    do = linear_dgrad(dp, mir(at.proj.weight), E, Q)
    tabs_p = _tables(at)
    dqkv, dtab = fused_attention.attn_bwd_raw(do.view(B, N, H, 64), qkv.view(B, N, 3, H, 64),
                                              *(t.detach() for t in tabs_p), o, lse, sp, scale, mr,
                                              reduce_tables=False)
    nb = tabs_p[0].shape[0]
    for i, t in enumerate(tabs_p):                                     # dtab (B*H, 4, 32, 64)
        jobs.add(t, dtab, dtab.shape[0], 4 * 32 * 64, nb, 64, src_offset=i * 32 * 64)
    dqkv2d = dqkv.view(M, 3 * Q)
    pwq = wgrad_parts_async(dqkv2d, a)                                       # rows [q | k | v]
    jobs.add(at.qkv.weight, pwq, pwq.shape[0], 3 * Q * E, 3 * Q, E, interleave=Q)
    pbq = colsum128(dqkv2d)
    jobs.add(at.qkv.bias, pbq, pbq.shape[0], 3 * Q, 1, 3 * Q)
    da = linear_dgrad(dqkv2d, wqkv, 3 * Q, E)
    dx, df_prev, pl1 = ln_bwd_raw(da, x, mean1, rstd1, ln1.weight[:E], dx1, prev_scale, N, want_prev)
    jobs.add(ln1.weight, pl1, P, 3 * E, 1, E)
    jobs.add(ln1.bias, pl1, P, 3 * E, 1, E, src_offset=E)
    finalize_on_side_stream(jobs, blk, [df, g, pw2, pb2[0], dh, c, pw1, pb1, pl2, dp, o, pwp, dtab, dqkv, a, pwq, pbq,
                                        pl1])
    return dx, df_prev, (pl1, P, 3 * E, 2 * E)


# ---- native sequencing: one C call per block and direction (csrc/block_seq.cpp) ---------------
NATIVE_BLOCK = True
_ws_cache = {}


def _ptr_or_null(t):
    return t.data_ptr() if t is not None else 0


def _ensure_grad(p):
    if p.grad is None:
        p.grad = torch.zeros_like(p, memory_format=torch.contiguous_format)
    return p.grad


# per-block caches live ON the module (blk.__dict__), so they die with it: keyed by id() they
# would be inherited by whatever module later gets the same address
_DESC_KEY, _GRADS_KEY = '_cream_desc_cache', '_cream_grads_cache'


def _block_params(blk):
    at = blk.attn
    return (at.qkv.weight, at.qkv.bias, at.proj.weight, at.proj.bias, blk.fc1.weight, blk.fc1.bias, blk.fc2.weight,
            blk.fc2.bias)


def _block_desc(blk, B, N):
    """cream_block_desc of this block for the sampled configuration.  The pointer part (bf16
    operand copies of the super weights, LayerNorm parameters, tables — all read in place) is
    built once per block and device; per call only the sampled extents are filled in."""
    at = blk.attn
    ent = blk.__dict__.get(_DESC_KEY)
    ln1, ln2 = blk.attn_layer_norm, blk.ffn_layer_norm
    key = (at.qkv.weight.data_ptr(), ln1.weight.data_ptr())          # moves when the module changes device / storage
    if ent is None or ent[1] != key:
        mir = MIRROR.get
        t = _lib.BlockDesc()
        ms = [mir(p) for p in _block_params(blk)]
        t.wqkv, t.bqkv, t.ld_qkv = ms[0].data_ptr(), ms[1].data_ptr(), ms[0].stride(0)
        t.wproj, t.bproj, t.ld_proj = ms[2].data_ptr(), ms[3].data_ptr(), ms[2].stride(0)
        t.w1, t.b1, t.ld_w1 = ms[4].data_ptr(), ms[5].data_ptr(), ms[4].stride(0)
        t.w2, t.b2, t.ld_w2 = ms[6].data_ptr(), ms[7].data_ptr(), ms[6].stride(0)
        t.ln1_g, t.ln1_b, t.ln2_g, t.ln2_b = (ln1.weight.data_ptr(), ln1.bias.data_ptr(), ln2.weight.data_ptr(),
                                              ln2.bias.data_ptr())
        tabs = _tables(at)
        t.tkv, t.tkh, t.tvv, t.tvh = (x.data_ptr() for x in tabs)
        t.ldt = tabs[0].stride(0)
        t.eps1, t.eps2 = ln1.eps, ln2.eps
        t.mr = at.max_relative_position
        ent = blk.__dict__[_DESC_KEY] = (t, key, ms)
    else:
        for p, m in zip(_block_params(blk), ent[2]):                 # operand copies still current?
            MIRROR.get(p)
    d = _lib.BlockDesc.from_buffer_copy(ent[0])
    d.B, d.N, d.E, d.H, d.F = B, N, blk.sample_embed_dim, at.sample_num_heads, blk.sample_ffn_embed_dim_this_layer
    d.gh, d.gw = fused_attention.grid_of(N, d.mr)
    d.wgrad_split = _wgrad_split(B * N)
    d.attn_scale = float(at.sample_scale)
    return d


def _block_grads(blk):
    """cream_block_grads: pointers of the fp32 gradient tensors (created zero-filled on first use;
    cached while the tensors stay the same objects — the reducer's flat-buffer views never move)."""
    at = blk.attn
    ln1, ln2 = blk.attn_layer_norm, blk.ffn_layer_norm
    ent = blk.__dict__.get(_GRADS_KEY)
    if ent is not None and all(p.grad is g for p, g in ent[1]):
        return ent[0]
    params = _block_params(blk) + (ln1.weight, ln1.bias, ln2.weight, ln2.bias) + _tables(at)
    gr = [_ensure_grad(p) for p in params]
    g = _lib.BlockGrads()
    g.wqkv, g.bqkv, g.wproj, g.bproj, g.w1, g.b1, g.w2, g.b2 = (t.data_ptr() for t in gr[:8])
    g.ld_qkv, g.ld_proj, g.ld_w1, g.ld_w2 = gr[0].stride(0), gr[2].stride(0), gr[4].stride(0), gr[6].stride(0)
    g.ln1_g, g.ln1_b, g.ln2_g, g.ln2_b = (t.data_ptr() for t in gr[8:12])
    g.tkv, g.tkh, g.tvv, g.tvh = (t.data_ptr() for t in gr[12:16])
    g.ldt = gr[12].stride(0)
    blk.__dict__[_GRADS_KEY] = (g, list(zip(params, gr)))
    return g


def _ws_layout(d):
    """(fwd bytes, off_x, off_x1, off_f, bwd bytes, off_dx, off_df_prev, off_pl1) of a configuration."""
    key = (d.B, d.N, d.E, d.H, d.F, d.wgrad_split)
    hit = _ws_cache.get(key)
    if hit is None:
        lib = _lib.load()
        o = [ctypes.c_int64() for _ in range(6)]
        ft = lib.cream_block_fwd_workspace(ctypes.byref(d), ctypes.byref(o[0]), ctypes.byref(o[1]), ctypes.byref(o[2]))
        bt = lib.cream_block_bwd_workspace(ctypes.byref(d), ctypes.byref(o[3]), ctypes.byref(o[4]), ctypes.byref(o[5]))
        if ft < 0 or bt < 0:
            raise RuntimeError("cream_amd: block workspace query failed")
        hit = _ws_cache[key] = (ft, o[0].value, o[1].value, o[2].value, bt, o[3].value, o[4].value, o[5].value)
    return hit


class StackFunction(torch.autograd.Function):
    """A run of supernet blocks as ONE autograd node: x (B, N, E) fp32 -> (B, N, E) fp32.
    scales: (L, 2, B) per-sample drop-path scales of the L blocks, or None.
    Besides saving autograd bookkeeping, keeping the blocks together lets the passes at a block
    boundary merge: the last residual add of block i rides on the first LayerNorm of block i+1,
    and that LayerNorm's backward emits block i's fc2-output gradient and fc2 bias gradient.
    With NATIVE_BLOCK every block is one call into the C ABI per direction (csrc/block_seq.cpp);
    otherwise the same kernels are driven op by op from here (`_block_forward/_block_backward`)."""

    @staticmethod
    def forward(ctx, x, scales, blks):
        if NATIVE_BLOCK:
            return StackFunction._forward_native(ctx, x, scales, blks)
        B, N, E = x.shape
        M = B * N
        cur = x.contiguous().view(M, E)
        pend = None
        saved, dims = [], []
        for i, blk in enumerate(blks):
            dp1 = scales[i, 0] if scales is not None else None
            dp2 = scales[i, 1] if scales is not None else None
            x1, f, d, sv = _block_forward(blk, cur, pend, dp1, B, N)
            saved.extend(sv)
            dims.append(d)
            cur, pend = x1, (f, dp2)
        out = residual_add(cur, pend[0], pend[1], N * E)
        ctx.native = False
        ctx.blks = list(blks)
        ctx.dims = dims
        ctx.nsaved = len(saved) // len(blks)
        ctx.has_scales = scales is not None
        ctx.save_for_backward(*saved, *([scales] if scales is not None else []))
        return out.view(B, N, E)

    @staticmethod
    def _forward_native(ctx, x, scales, blks):
        B, N, E = x.shape
        lib = _lib.load()
        x = x.contiguous()
        dev = x.device
        stream = _gemm_stream(dev)
        sc_ptr = scales.data_ptr() if scales is not None else 0
        cur, pend_f, pend_s = x.data_ptr(), 0, 0
        descs, wss, xptrs = [], [], []
        for i, blk in enumerate(blks):
            d = _block_desc(blk, B, N)
            ft, off_x, off_x1, off_f = _ws_layout(d)[:4]
            ws = torch.empty(ft, dtype=torch.uint8, device=dev)
            wp = ws.data_ptr()
            dp1 = sc_ptr + (2 * i) * B * 4 if sc_ptr else 0
            _lib.check(lib.cream_block_fwd(ctypes.byref(d), wp, cur, pend_f, pend_s, dp1, stream), "cream_block_fwd")
            xptrs.append(wp + off_x if pend_f else cur)
            descs.append(d)
            wss.append(ws)
            cur, pend_f, pend_s = wp + off_x1, wp + off_f, (sc_ptr + (2 * i + 1) * B * 4 if sc_ptr else 0)
        out = torch.empty((B, N, E), dtype=torch.float32, device=dev)
        _lib.check(lib.cream_residual_add(out.data_ptr(), cur, pend_f, pend_s, B * N * E, N * E, stream), "cream_residual_add")
        ctx.native = True
        ctx.blks = list(blks)
        ctx.descs = descs
        ctx.xptrs = xptrs
        ctx.shape = (B, N, E)
        ctx.has_scales = scales is not None
        ctx.save_for_backward(x, *wss, *([scales] if scales is not None else []))
        return out

    @staticmethod
    def backward(ctx, dout):
        if ctx.native:
            return StackFunction._backward_native(ctx, dout)
        blks = ctx.blks
        tens = ctx.saved_tensors
        scales = tens[-1] if ctx.has_scales else None
        ns = ctx.nsaved
        B, N, E = ctx.dims[0][:3]
        dx = dout.contiguous().view(B * N, E)
        L = len(blks)
        # gradient of the last block's fc2 output: s2 * dout (+ its column sums)
        df, part = scale_cast_colsum(dx, scales[L - 1, 1] if scales is not None else None, N)
        pb2 = (part, part.shape[0], E, 0)
        for i in range(L - 1, -1, -1):
            dp1 = scales[i, 0] if scales is not None else None
            prev_scale = scales[i - 1, 1] if (scales is not None and i > 0) else None
            dx, df, pb2 = _block_backward(blks[i], ctx.dims[i], dp1, tens[i * ns:(i + 1) * ns], dx, df, pb2,
                                          prev_scale, i > 0)
        join_side_stream(dx.device)           # every parameter gradient of the run is complete
        return dx.view(B, N, E), None, None

    @staticmethod
    def _backward_native(ctx, dout):
        blks = ctx.blks
        tens = ctx.saved_tensors
        scales = tens[-1] if ctx.has_scales else None
        wss = tens[1:1 + len(blks)]
        B, N, E = ctx.shape
        M = B * N
        lib = _lib.load()
        dev = dout.device
        stream = _gemm_stream(dev)
        if WGRAD_SIDE_STREAM:
            side_st = _side_stream(dev)
            with torch.cuda.stream(side_st):
                side = _gemm_stream(dev)
        else:
            side_st, side = None, stream
        sc_ptr = scales.data_ptr() if scales is not None else 0
        L = len(blks)
        dx_t = dout.contiguous().view(M, E)
        df_t, part = scale_cast_colsum(dx_t, scales[L - 1, 1] if scales is not None else None, N)
        keep = [dx_t, df_t, part]
        dx, df, pb2, pb2_parts, pb2_stride = dx_t.data_ptr(), df_t.data_ptr(), part.data_ptr(), part.shape[0], E
        ws = None
        off_dx = 0
        for i in range(L - 1, -1, -1):
            d = ctx.descs[i]
            bt, off_dx, off_dfp, off_pl1 = _ws_layout(d)[4:]
            ws = torch.empty(bt, dtype=torch.uint8, device=dev)
            keep.append(ws)
            wp = ws.data_ptr()
            g = _block_grads(blks[i])
            dp1 = sc_ptr + (2 * i) * B * 4 if sc_ptr else 0
            prev = sc_ptr + (2 * (i - 1) + 1) * B * 4 if (sc_ptr and i > 0) else 0
            _lib.check(lib.cream_block_bwd(ctypes.byref(d), ctypes.byref(g), wss[i].data_ptr(), ctx.xptrs[i], wp, dx, df, pb2,
                                           pb2_parts, pb2_stride, dp1, prev, 1 if i > 0 else 0, stream, side),
                       "cream_block_bwd")
            if _grad_ready_hooks:
                if side_st is not None:
                    with torch.cuda.stream(side_st):
                        _notify(blks[i])
                else:
                    _notify(blks[i])
            dx, df = wp + off_dx, wp + off_dfp
            pb2, pb2_parts, pb2_stride = wp + off_pl1 + 2 * E * 4, lib.cream_ln_partials(), 3 * E
        join_side_stream(dev)                 # every parameter gradient of the run is complete
        out = ws[off_dx:off_dx + M * E * 4].view(torch.float32).view(B, N, E)
        return out, None, None


class BlockFunction:
    """One block = a stack of one.  apply(x, dp1, dp2, blk); dp1 / dp2: per-sample drop-path
    scales (B,) or None."""

    @staticmethod
    def apply(x, dp1, dp2, blk):
        scales = None
        if dp1 is not None or dp2 is not None:
            ones = torch.ones(x.shape[0], device=x.device, dtype=torch.float32)
            scales = torch.stack([dp1 if dp1 is not None else ones, dp2 if dp2 is not None else ones]).unsqueeze(0)
        return StackFunction.apply(x, scales, [blk])

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
  * the dense projections: hand-written MFMA GEMMs (csrc/gemm_mfma.hpp) on bf16 OPERAND COPIES of
    the fp32 master weights, read in place as `W[:out, :in]` (leading dimension = super width):
    forward on W (qkv on its de-interleaved [q | k | v] parts), dgrad on the transposed copies,
    fc1 with the erf-GELU and fc2's dgrad with GELU' in their epilogues; weight gradients are
    split-K products over the 25k tokens (transpose-reads of both operands) whose fp32 partials
    `cream_grad_finalize` adds into the active slice of the fp32 `.grad`.  The copies are written
    by the optimizer kernel (csrc/optim.hip) in the same pass that updates the weights.
Two drivers over the SAME kernels: `NATIVE_BLOCK` (default) = one call into the C ABI per block and
direction (csrc/block_seq.cpp; weight gradients + finalisation on a side stream), or op by op from
here (`_block_forward` / `_block_backward`; used by the kernel-timing pass of bench.py and by the
tests that pin the native sequencing bit for bit).  The backward is written by hand; parameter
gradients are accumulated into `p.grad` directly and announced through the `on_grads_ready` hooks
(the gradient reducer starts a block's all-reduce the moment its last gradient exists).
"""
import ctypes
import os

import torch

from .. import _lib, timing
from . import fused_attention

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


# ---- dense projections: hand-written MFMA GEMMs (csrc/gemm_mfma.hip) -------------------------------
def linear_fwd(x, w, bias, N, K, out=None):
    """out (M, N) = x (M, K) . W[:N, :K]^T + bias[:N]; w: bf16 (rows, ld) operand copy read in place."""
    M = x.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=torch.bfloat16, device=x.device)
    with timing.region("gemm_nt", flops=2 * M * N * K):
        _lib.check(_lib.load().cream_linear_fwd(_p(out), _p(x), _p(w), _p(bias), M, N, K, w.stride(0), _stream()),
                   "cream_linear_fwd")
    return out


def linear_fwd_seg(x, w3, bias, N, K, nseg, out=None):
    """The same with W = the first `nseg` rows of each of the 3 parts of w3 (3, rows, ld): the
    de-interleaved qkv operand; N = 3 * nseg."""
    M = x.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=torch.bfloat16, device=x.device)
    with timing.region("gemm_nt", flops=2 * M * N * K):
        _lib.check(_lib.load().cream_linear_fwd_seg(_p(out), _p(x), _p(w3), _p(bias), M, N, K, w3.stride(1), nseg,
                                                    w3.stride(0), _stream()), "cream_linear_fwd_seg")
    return out


def linear_gelu_fwd(x, w, bias, N, K, want_grad=True):
    """-> (gp, g) with h = bf16(x . W^T + bias): gp = gelu'(float(h)), g = gelu(float(h)), both bf16, in
    one pass (fc1 + activation; gp is everything the backward needs of h).  want_grad=False: gp is None and is
    not written (forward without a backward)."""
    M = x.shape[0]
    g = torch.empty((M, N), dtype=torch.bfloat16, device=x.device)
    h = torch.empty_like(g) if want_grad else None
    with timing.region("gemm_nt_gelu", flops=2 * M * N * K):
        _lib.check(_lib.load().cream_linear_gelu_fwd(_p(h), _p(g), _p(x), _p(w), _p(bias), M, N, K, w.stride(0), _stream()),
                   "cream_linear_gelu_fwd")
    return h, g


def linear_dgrad(dy, wt, N, K, out=None):
    """dx (M, K) = dy (M, N) . W[:N, :K]; wt: the TRANSPOSED operand copy (in, out)."""
    M = dy.shape[0]
    if out is None:
        out = torch.empty((M, K), dtype=torch.bfloat16, device=dy.device)
    with timing.region("gemm_nt", flops=2 * M * N * K):
        _lib.check(_lib.load().cream_linear_dgrad(_p(out), _p(dy), _p(wt), M, N, K, wt.stride(0), _stream()),
                   "cream_linear_dgrad")
    return out


def linear_dgrad_seg(dy, wt3, N, K, kseg, out=None):
    """dx = dy . W with W's rows = the first `kseg` rows of the 3 qkv parts; wt3 (3, in, out)."""
    M = dy.shape[0]
    if out is None:
        out = torch.empty((M, K), dtype=torch.bfloat16, device=dy.device)
    with timing.region("gemm_nt", flops=2 * M * N * K):
        _lib.check(_lib.load().cream_linear_dgrad_seg(_p(out), _p(dy), _p(wt3), M, N, K, wt3.stride(1), kseg, wt3.stride(0),
                                                      _stream()), "cream_linear_dgrad_seg")
    return out


def linear_dgrad_mul(dy, wt, factor, N, K):
    """-> (dh, parts): dh (M, K) = (dy . W) * factor (the saved gelu'(h)), parts (slabs, K) its per-slab
    column sums."""
    M = dy.shape[0]
    lib = _lib.load()
    dh = torch.empty((M, K), dtype=torch.bfloat16, device=dy.device)
    parts = torch.empty((lib.cream_colsum128_slabs(M), K), dtype=torch.float32, device=dy.device)
    with timing.region("gemm_nt_mul", flops=2 * M * N * K):
        _lib.check(lib.cream_linear_dgrad_mul(_p(dh), _p(parts), _p(dy), _p(wt), _p(factor), M, N, K, wt.stride(0), _stream()),
                   "cream_linear_dgrad_mul")
    return dh, parts


def wgrad_parts_dtype():
    """bf16 when the library writes the split-K partial tiles of the block's weight gradients as bf16
    (cream_block_wgrad_bf16: half the partial traffic), else fp32 — the op-by-op path follows the native one."""
    return torch.bfloat16 if _lib.load().cream_block_wgrad_bf16(-1) else torch.float32


def _wgrad_splits(M, N, K, want_bias, parts_dtype):
    """token slices of a weight gradient: bf16 partial tiles take the split count of the macro-tile kernel
    (cream_linear_wgrad_splits_bf16, csrc/gemm_tn8.hpp) — the library picks the kernel by the S it is called with."""
    lib = _lib.load()
    if parts_dtype == torch.bfloat16:
        return lib.cream_linear_wgrad_splits_bf16(M, N, K)
    return lib.cream_linear_wgrad_splits(M, N, K)


def linear_wgrad_parts(dy, x, want_bias=False, out=None, bias_out=None, parts_dtype=torch.float32):
    """-> (parts (S, N, K) fp32 or bf16, bias_parts (S, N) fp32 or None): partial products dy_s^T x_s over S
    slices of the token dimension (S chosen by the library) and, on request, the column sums of dy_s."""
    M, N = dy.shape
    K = x.shape[1]
    lib = _lib.load()
    if out is None:
        S = _wgrad_splits(M, N, K, want_bias, parts_dtype)
        out = torch.empty((S, N, K), dtype=parts_dtype, device=dy.device)
    S = out.shape[0]
    if want_bias and bias_out is None:
        bias_out = torch.empty((S, N), dtype=torch.float32, device=dy.device)
    fn, name = ((lib.cream_linear_wgrad_parts_bf16, "cream_linear_wgrad_parts_bf16") if out.dtype == torch.bfloat16
                else (lib.cream_linear_wgrad_parts, "cream_linear_wgrad_parts"))
    with timing.region("gemm_tn_wgrad", flops=2 * M * N * K):
        _lib.check(fn(_p(out), _p(bias_out) if want_bias else ctypes.c_void_p(0), _p(dy), _p(x), M, N, K, S, _stream()), name)
    return out, (bias_out if want_bias else None)


# The weight-gradient GEMMs are off the critical path of a block's backward (only the gradient
# finalisation at its end reads them) and MFMA-bound, while the passes between the dgrad GEMMs
# (GELU', LayerNorm', column sums) are HBM-bound: enqueued on a second HIP stream they fill the
# matrix cores while the main stream streams memory, and the tails of the small-N GEMMs overlap.
_side_streams = {}
WGRAD_SIDE_STREAM = True


def _side_stream(device):
    st = _side_streams.get(device)
    if st is None:
        # (stream priorities were tried — side stream at 0 / main at -1 and the reverse: no change in step time;
        # a CU-masked side stream (hipExtStreamCreateWithCUMask, 64-128 CUs, whole XCDs or spread): 10.8 -> 16.2 ms)
        st = _side_streams[device] = torch.cuda.Stream(device=device)
    return st


def wgrad_parts_async(dy, x, want_bias=False):
    """linear_wgrad_parts on the side stream: the outputs are allocated on the caller's stream (so the
    caching allocator never hands them out while the side stream still writes them: the caller joins
    the side stream before the partials are consumed), the operands are complete at this point of
    the caller's stream (event)."""
    if not WGRAD_SIDE_STREAM:
        return linear_wgrad_parts(dy, x, want_bias, parts_dtype=wgrad_parts_dtype())
    M, N = dy.shape
    S = _wgrad_splits(M, N, x.shape[1], want_bias, wgrad_parts_dtype())
    out = torch.empty((S, N, x.shape[1]), dtype=wgrad_parts_dtype(), device=dy.device)
    bout = torch.empty((S, N), dtype=torch.float32, device=dy.device) if want_bias else None
    main = torch.cuda.current_stream(dy.device)
    side = _side_stream(dy.device)
    side.wait_event(main.record_event())
    with torch.cuda.stream(side):
        linear_wgrad_parts(dy, x, want_bias, out=out, bias_out=bout)
    return out, bout


def join_side_stream(device):
    if WGRAD_SIDE_STREAM:
        torch.cuda.current_stream(device).wait_stream(_side_stream(device))


# The main stream has to wait for the weight-gradient stream once per backward pass.  Done right behind the first block's backward
# the wait exposes the side stream's tail (the first block's qkv weight gradient and finalisation start when the main chain is
# almost through: ~50 us per step in the kernel trace, profiles/r05_step_tail.md) while the stem's backward — which needs nothing from
# the side stream — queues up behind it.  Deferred to autograd's end-of-pass callback (the hook DistributedDataParallel finalises
# its buckets from) the stem's backward runs under that tail.  CREAM_DEFER_JOIN=0: wait where the blocks end.
DEFER_JOIN = os.environ.get('CREAM_DEFER_JOIN', '1') != '0'


def join_side_stream_at_end_of_backward(device, held=()):
    """`held`: tensors the side stream may still be reading when the caller returns (allocated on the main stream).  The callback
    keeps them referenced until the join is on the main stream, so the caching allocator cannot hand them to the kernels that now
    run before it (record_stream would do, but it defers the reuse to an event query at the next allocation: measured 5-20 %
    slower per step).  One callback per call, no state: a backward pass that raises leaves nothing behind."""
    if not WGRAD_SIDE_STREAM:
        return
    if not DEFER_JOIN:
        join_side_stream(device)
        return
    held = list(held)
    # the stream the backward of the blocks RUNS on (this call is made from StackFunction.backward): the callback may be run by
    # another thread / under another current stream, and the workspaces in `held` belong to this stream's allocator pool
    main = torch.cuda.current_stream(device)
    side = _side_stream(device)

    def _join():
        main.wait_stream(side)
        cur = torch.cuda.current_stream(device)
        if cur != main:
            cur.wait_stream(side)
        held.clear()

    try:
        torch.autograd.Variable._execution_engine.queue_callback(_join)
    except RuntimeError:                     # not inside an engine pass (backward driven by hand): join here
        _join()


def finalize_on_side_stream(jobs, blk, tensors):
    """The finalisation of the block's gradients and their announcement to the reducer, on the side stream: the main
    stream goes straight on to the previous block.  `tensors`: everything the side stream reads — they were allocated on
    the main stream, so the allocator is told not to recycle them before the side stream is done."""
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

    def add(self, param, src, nparts, pstride, rows, cols, interleave=0, src_offset=0, ld=None):
        fresh = False
        if param.grad is None:
            # a job that covers the whole parameter writes its gradient instead of adding to a zero fill (one launch less per
            # parameter and step where the optimizer drops the gradients: TinyCLIP / DeiT towers); slices of a super-weight
            # (AutoFormer) need the zeros around them
            fresh = not interleave and ld is None and rows * cols == param.numel()
            param.grad = (torch.empty_like if fresh else torch.zeros_like)(param, memory_format=torch.contiguous_format)
        g = param.grad
        assert g.is_contiguous() or g.dim() <= 2
        j = self.jobs[self.n]
        j.overwrite = 1 if fresh else 0
        j.dst = g.data_ptr()
        j.src = src.data_ptr() + src_offset * src.element_size()
        j.ld = ld if ld is not None else (g.stride(0) if g.dim() == 2 else cols)
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
    """dW (out, in) fp32 = dy^T x (split-K partials added here; the block path adds them in
    cream_grad_finalize)."""
    return linear_wgrad_parts(dy, x)[0].sum(dim=0)


# ---- bf16 operand copies of the fp32 master weights -----------------------------------------------
def param_job(p, grad=None, exp_avg=None, exp_avg_sq=None, mir=None, mir_t=None, deinterleave=False, weight_decay=0.0):
    """One cream_param_job: the parameter viewed as (rows, cols) = (numel / last dim, last dim)."""
    cols = p.shape[-1] if p.dim() > 1 else p.numel()
    rows = p.numel() // cols
    assert p.is_contiguous()
    j = _lib.ParamJob()
    j.p = p.data_ptr()
    j.g = grad.data_ptr() if grad is not None else 0
    j.m = exp_avg.data_ptr() if exp_avg is not None else 0
    j.v = exp_avg_sq.data_ptr() if exp_avg_sq is not None else 0
    j.ld, j.rows, j.cols = cols, rows, cols
    j.deinterleave = 3 if deinterleave else 0
    j.weight_decay = weight_decay
    if mir is not None:
        j.mir = mir.data_ptr()
        j.ld_mir = mir.stride(-2) if mir.dim() >= 2 else cols
        j.seg_stride = mir.stride(0) if deinterleave else 0
    if mir_t is not None:
        j.mir_t = mir_t.data_ptr()
        j.ld_mir_t = mir_t.stride(-2)
        j.seg_stride_t = mir_t.stride(0) if deinterleave else 0
    return j


class JobTable:
    """A device-resident table of cream_param_job + the prefix sums of their tile counts (the host
    builds it once: the pointers of parameters, gradients, moments and copies never move)."""

    def __init__(self, jobs, device):
        import numpy as np
        lib = _lib.load()
        self.n = len(jobs)
        arr = (_lib.ParamJob * self.n)(*jobs)
        first = [0]
        for j in jobs:
            first.append(first[-1] + lib.cream_param_job_tiles(j.rows, j.cols))
        self.total = first[-1]
        self.jobs = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)
        self.first = torch.tensor(first, dtype=torch.int32).to(device)

    def launch(self, update=False, lr=0.0, beta1=0.9, beta2=0.999, eps=1e-8, step=1):
        _lib.check(_lib.load().cream_adamw_step(_p(self.jobs), _p(self.first), self.n, self.total, 1 if update else 0,
                                                lr, beta1, beta2, eps, step, _stream()), "cream_adamw_step")


class BlockOperands:
    """bf16 operand copies of one block's projection weights in the layouts csrc/gemm_mfma.hip reads:
    W (out, in) and W^T (in, out) with the super widths as leading dimensions; qkv de-interleaved into
    its three parts (3, Qmax, in) / (3, in, Qmax); biases as plain bf16 vectors."""

    NAMES = ('qkv', 'proj', 'fc1', 'fc2')

    def __init__(self, blk):
        at = blk.attn
        self.mods = (at.qkv, at.proj, blk.fc1, blk.fc2)
        dev = at.qkv.weight.device
        bf = dict(dtype=torch.bfloat16, device=dev)
        self.w, self.wt, self.b = [], [], []
        for name, m in zip(self.NAMES, self.mods):
            out_f, in_f = m.weight.shape
            if name == 'qkv':
                self.w.append(torch.empty((3, out_f // 3, in_f), **bf))
                self.wt.append(torch.empty((3, in_f, out_f // 3), **bf))
            else:
                self.w.append(torch.empty((out_f, in_f), **bf))
                self.wt.append(torch.empty((in_f, out_f), **bf))
            self.b.append(torch.empty((out_f,), **bf) if m.bias is not None else None)
        # bf16 operand images of the four relative-position tables (include/cream_amd.h: cream_attn_rpe2d_table_images layout):
        # rows and transposed copies are exactly what cream_adamw_step writes for any parameter (mir / mir_t), so the images
        # are kept current by the optimizer kernel like the GEMM operand copies — the attention forward then runs the
        # ping-pong kernel and the backward saves its image launch
        self.tables, self.timg, self._timg_views = (), None, ()
        if getattr(at, 'relative_position', False):
            tabs = _tables(at)
            nb = tabs[0].shape[0]
            if all(t.shape == (nb, 64) and t.is_contiguous() and t.dtype == torch.float32 for t in tabs) and nb <= 32:
                self.tables = tabs
                self.timg = torch.zeros((_lib.load().cream_attn_rpe2d_table_image_bytes() // 2,), **bf)
                views = []
                for i in range(4):
                    base = (i // 2) * 8192 + (i % 2) * 32 * 64
                    base_t = (i // 2) * 8192 + 4096 + (i % 2) * 32
                    views.append((self.timg[base:base + nb * 64].view(nb, 64),
                                  torch.as_strided(self.timg, (64, nb), (64, 1), base_t)))
                self._timg_views = tuple(views)
        self.key = self._key()
        self.versions = None
        self._table = None
        _register(self)

    def _key(self):
        return tuple(m.weight.data_ptr() for m in self.mods) + tuple(t.data_ptr() for t in self.tables)

    def _params(self):
        return [p for m in self.mods for p in (m.weight, m.bias) if p is not None] + list(self.tables)

    def jobs(self, grads=False, states=None, weight_decay=0.0):
        """cream_param_jobs of this block's 8 projection tensors (weights with both copies, biases with
        their bf16 copy).  With `states` (parameter -> (exp_avg, exp_avg_sq)) they are update jobs."""
        out = []
        for name, m, w, wt, b in zip(self.NAMES, self.mods, self.w, self.wt, self.b):
            st = states[m.weight] if states else (None, None)
            out.append((m.weight, param_job(m.weight.detach(), m.weight.grad if grads else None, st[0], st[1], w, wt,
                                            deinterleave=(name == 'qkv'), weight_decay=weight_decay)))
            if m.bias is not None:
                st = states[m.bias] if states else (None, None)
                out.append((m.bias, param_job(m.bias.detach(), m.bias.grad if grads else None, st[0], st[1], b, None)))
        for t, (rows, rows_t) in zip(self.tables, self._timg_views):
            st = states[t] if states else (None, None)
            out.append((t, param_job(t.detach(), t.grad if grads else None, st[0], st[1], rows, rows_t)))
        return out

    def refresh(self):
        if self._table is None:
            self._table = JobTable([j for _, j in self.jobs()], self.w[0].device)
        self._table.launch(update=False)
        self.mark_fresh()

    def mark_fresh(self):
        self.versions = tuple(p._version for p in self._params())

    def stale(self):
        return self.versions != tuple(p._version for p in self._params())


_OPS_KEY = '_cream_operands'

# Every live operand set, so that an optimizer the library does not know can invalidate them: fused optimizers
# (torch.optim.AdamW / SGD with fused=True) and writes through `.data` leave `Tensor._version` unchanged, so the version
# test of `stale()` alone would keep the block GEMMs on the initial weights forever.  engine.NativeAdamW rewrites the
# copies itself and is skipped.
import weakref as _weakref

_ALL_OPERANDS = _weakref.WeakSet()
_optimizer_hook = None


def _register(ops):
    global _optimizer_hook
    _ALL_OPERANDS.add(ops)
    if _optimizer_hook is None:
        from torch.optim.optimizer import register_optimizer_step_post_hook

        def after_any_optimizer_step(opt, args, kwargs):
            if getattr(opt, '_cream_rewrites_operands', False):
                return
            mine = {p.data_ptr() for g in opt.param_groups for p in g['params']}
            for ops in list(_ALL_OPERANDS):
                if any(k in mine for k in ops._key()):
                    ops.versions = None                      # stale: re-derived on the next use

        _optimizer_hook = register_optimizer_step_post_hook(after_any_optimizer_step)


class PatchOperands:
    """bf16 operand copy of the patch-embedding weight viewed as (E_super, C*ph*pw) and of its bias — the
    stem's projection runs on the same NT / TN GEMMs as the block projections (embedding_super.py:27-40 is
    a stride = kernel convolution, i.e. a GEMM over unfolded patches)."""

    def __init__(self, pe):
        self.mod = pe.proj
        wt = pe.proj.weight
        bf = dict(dtype=torch.bfloat16, device=wt.device)
        self.w = torch.empty((wt.shape[0], wt.numel() // wt.shape[0]), **bf)
        self.b = torch.empty((wt.shape[0],), **bf)
        self.key = self._key()
        self.versions = None
        self._table = None
        _register(self)

    def _key(self):
        return (self.mod.weight.data_ptr(),)

    def jobs(self, grads=False, states=None, weight_decay=0.0):
        m = self.mod
        out = []
        for prm, mir in ((m.weight, self.w), (m.bias, self.b)):
            st = states[prm] if states else (None, None)
            view = prm.detach().view(self.w.shape) if prm is m.weight else prm.detach()
            grad = (prm.grad.view(self.w.shape) if prm is m.weight else prm.grad) if grads else None
            stv = tuple(t.view(view.shape) if t is not None else None for t in st)
            out.append((prm, param_job(view, grad, stv[0], stv[1], mir, None, weight_decay=weight_decay)))
        return out

    def refresh(self):
        if self._table is None:
            self._table = JobTable([j for _, j in self.jobs()], self.w.device)
        self._table.launch(update=False)
        self.mark_fresh()

    def mark_fresh(self):
        self.versions = (self.mod.weight._version, self.mod.bias._version)

    def stale(self):
        return self.versions != (self.mod.weight._version, self.mod.bias._version)


def patch_operands(pe, fresh=True):
    ops = pe.__dict__.get(_OPS_KEY)
    if ops is None or ops.key != ops._key():
        ops = pe.__dict__[_OPS_KEY] = PatchOperands(pe)
    if fresh and ops.stale():
        ops.refresh()
    return ops


class HeadOperands:
    """bf16 operand copies of the classifier (LinearSuper `head`, supernet_transformer.py:136): W (classes, E_super), its
    transpose and the bias — the head runs on the same NT / TN GEMMs as the block projections."""

    def __init__(self, head):
        self.mod = head
        w = head.weight
        bf = dict(dtype=torch.bfloat16, device=w.device)
        self.w = torch.empty(tuple(w.shape), **bf)
        self.wt = torch.empty((w.shape[1], w.shape[0]), **bf)
        self.b = torch.empty((w.shape[0],), **bf)
        self.key = self._key()
        self.versions = None
        self._table = None
        _register(self)

    def _key(self):
        return (self.mod.weight.data_ptr(),)

    def jobs(self, grads=False, states=None, weight_decay=0.0):
        m = self.mod
        out = []
        for prm, mir, mir_t in ((m.weight, self.w, self.wt), (m.bias, self.b, None)):
            st = states[prm] if states else (None, None)
            out.append((prm, param_job(prm.detach(), prm.grad if grads else None, st[0], st[1], mir, mir_t, weight_decay=weight_decay)))
        return out

    def refresh(self):
        if self._table is None:
            self._table = JobTable([j for _, j in self.jobs()], self.w.device)
        self._table.launch(update=False)
        self.mark_fresh()

    def mark_fresh(self):
        self.versions = (self.mod.weight._version, self.mod.bias._version)

    def stale(self):
        return self.versions != (self.mod.weight._version, self.mod.bias._version)


def head_operands(head, fresh=True):
    ops = head.__dict__.get(_OPS_KEY)
    if ops is None or ops.key != ops._key():
        ops = head.__dict__[_OPS_KEY] = HeadOperands(head)
    if fresh and ops.stale():
        ops.refresh()
    return ops


def _no_frozen_parameters(mod):
    """The native backward of a node writes the gradient of EVERY parameter of that node into `.grad`: a node with
    frozen parameters that is differentiated through (partial fine-tuning) takes the composed path instead."""
    if not torch.is_grad_enabled():
        return True
    return all(p.requires_grad for p in mod.parameters(recurse=True))


def head_supported(head, feat):
    return (NATIVE_ENDS_GRADS and feat.is_cuda and feat.dim() == 2 and feat.dtype == torch.float32
            and torch.is_autocast_enabled('cuda') and torch.get_autocast_dtype('cuda') == torch.bfloat16
            and getattr(head, 'bias', None) is not None and not getattr(head, 'scale', False)
            and head.weight.dtype == torch.float32 and head.sample_in_dim % 8 == 0 and head.sample_out_dim % 8 == 0
            and head.sample_in_dim == feat.shape[1] and _no_frozen_parameters(head))


# Parameter gradients of the two ends of forward_features and of the classifier are ADDED INTO p.grad by one
# cream_grad_finalize launch per node (fixed summation order) and announced through the on_grads_ready hooks, like the
# blocks' — instead of zero-filled super-shaped temporaries + reductions + autograd's accumulate (round 2: ~45 fills and
# 19 copies per step in this path).
NATIVE_ENDS_GRADS = os.environ.get('CREAM_NATIVE_ENDS_GRADS', '1') != '0'


def _accumulate_and_notify(jobs, params):
    jobs.launch()
    if _grad_ready_hooks:
        for fn in _grad_ready_hooks:
            fn(params)


class HeadFunction(torch.autograd.Function):
    """logits (B, classes) bf16 = feat (B, E) . W[:, :E]^T + b on the own NT GEMM (LinearSuper.forward of the classifier,
    Linear_super.py:51-54 under autocast); backward: dgrad on the transposed copy, weight + bias gradient in one TN launch
    whose partials go straight into weight.grad[:, :E] / bias.grad."""

    @staticmethod
    def forward(ctx, feat, weight, bias, head):
        ops = head_operands(head)
        C, E = head.sample_out_dim, head.sample_in_dim
        x = feat.to(torch.bfloat16).contiguous()
        logits = linear_fwd(x, ops.w, ops.b, C, E)
        ctx.save_for_backward(x)
        ctx.head, ctx.dims = head, (C, E)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        (x,) = ctx.saved_tensors
        head = ctx.head
        C, E = ctx.dims
        ops = head_operands(head, fresh=False)
        dl = dlogits if dlogits.dtype == torch.bfloat16 else dlogits.to(torch.bfloat16)
        dl = dl.contiguous()
        dfeat = linear_dgrad(dl, ops.wt, C, E).float()
        parts, bparts = linear_wgrad_parts(dl, x, want_bias=True)
        jobs = GradJobs()
        jobs.add(head.weight, parts, parts.shape[0], C * E, C, E)
        jobs.add(head.bias, bparts, bparts.shape[0], C, 1, C)
        _accumulate_and_notify(jobs, [head.weight, head.bias])
        return dfeat, None, None, None


def head(head_mod, feat):
    return HeadFunction.apply(feat, head_mod.weight, head_mod.bias, head_mod)


class SoftTargetCEFunction(torch.autograd.Function):
    """mean_b sum_c -t log_softmax(x) with the logit gradient produced by the SAME launch (cream_soft_ce)."""

    @staticmethod
    def forward(ctx, logits, target):
        lib = _lib.load()
        B, C = logits.shape
        x = logits.contiguous()
        t = target.contiguous()
        rows = torch.empty(B, dtype=torch.float32, device=x.device)
        dl = torch.empty((B, C), dtype=torch.float32, device=x.device)
        code = _lib.BF16 if x.dtype == torch.bfloat16 else _lib.F32
        _lib.check(lib.cream_soft_ce(_p(rows), _p(dl), _p(x), _p(t), B, C, code, 1.0 / B, _stream()), "cream_soft_ce")
        ctx.save_for_backward(dl)
        ctx.out_dtype = logits.dtype
        return rows.mean()

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        out = torch.empty(dl.shape, dtype=ctx.out_dtype, device=dl.device)
        torch.mul(dl, g, out=out)                                # one rounding to the logits' dtype, as at the autocast boundary
        return out, None


def soft_ce_supported(logits, target):
    return (logits.is_cuda and logits.dim() == 2 and logits.shape[1] <= 2048 and logits.dtype in (torch.bfloat16, torch.float32)
            and target.dtype == torch.float32 and target.shape == logits.shape and not target.requires_grad)


class PatchEmbedFunction(torch.autograd.Function):
    """y (B*P, E) = patches (B*P, K) . W[:E]^T + b[:E] on the own GEMMs; backward = the split-K TN product
    (weight and bias gradient in one launch) added into the active rows of the conv weight's gradient.  The
    images need no gradient."""

    @staticmethod
    def forward(ctx, patches, weight, bias, pe, E):
        ops = patch_operands(pe)
        K = ops.w.shape[1]
        y = linear_fwd(patches, ops.w, ops.b, E, K)
        ctx.save_for_backward(patches)
        ctx.pe, ctx.E, ctx.K = pe, E, K
        return y

    @staticmethod
    def backward(ctx, dy):
        (patches,) = ctx.saved_tensors
        pe, E, K = ctx.pe, ctx.E, ctx.K
        dy = dy.to(torch.bfloat16).contiguous()
        parts, bparts = linear_wgrad_parts(dy, patches, want_bias=True)
        w = pe.proj.weight
        gw = parts.sum(0).view(E, *w.shape[1:])
        gb = bparts.sum(0)
        return None, _pad_rows(gw, w.shape[0]), _pad_rows(gb, w.shape[0]), None, None


def _pad_rows(g, rows):
    if g.shape[0] == rows:
        return g
    out = g.new_zeros((rows,) + tuple(g.shape[1:]))
    out[:g.shape[0]] = g
    return out


class StemFunction(torch.autograd.Function):
    """img (B, C, H, W) fp32 -> x0 (B, N, E) fp32, the input of the first block: patch embedding on the own
    GEMMs + class token + position embedding (supernet_transformer.py:147-155) in three launches
    (cream_im2patch, NT GEMM, cream_stem_assemble); backward in three (cream_stem_bwd, TN GEMM, one sum)."""

    @staticmethod
    def forward(ctx, img, weight, bias, cls, pos, pe, E):
        lib = _lib.load()
        B, C, H, W = img.shape
        ph, pw = pe.patch_size
        P = (H // ph) * (W // pw)
        N = P + 1
        dev = img.device
        ops = patch_operands(pe)
        K = ops.w.shape[1]
        img = img.contiguous()
        patches = torch.empty((B * P, K), dtype=torch.bfloat16, device=dev)
        stream = _stream()
        _lib.check(lib.cream_im2patch(_p(patches), _p(img), B, C, H, W, ph, pw, stream), "cream_im2patch")
        y = linear_fwd(patches, ops.w, ops.b, E, K)
        x0 = torch.empty((B, N, E), dtype=torch.float32, device=dev)
        cls_e = cls.detach().reshape(-1)[:E].contiguous()
        _lib.check(lib.cream_stem_assemble(_p(x0), _p(y), _p(cls_e), _p(pos) if pos is not None else ctypes.c_void_p(0),
                                           pos.shape[-1] if pos is not None else 0, B, N, E, stream), "cream_stem_assemble")
        ctx.save_for_backward(patches)
        ctx.pe, ctx.E, ctx.K, ctx.dims = pe, E, K, (B, N)
        ctx.params = (weight, bias, cls, pos)
        ctx.shapes = (tuple(weight.shape), tuple(bias.shape), tuple(cls.shape), tuple(pos.shape) if pos is not None else None)
        return x0

    @staticmethod
    def backward(ctx, dx0):
        (patches,) = ctx.saved_tensors
        lib = _lib.load()
        E, K = ctx.E, ctx.K
        B, N = ctx.dims
        dev = dx0.device
        dx0 = dx0.contiguous()
        dy = torch.empty((B * (N - 1), E), dtype=torch.bfloat16, device=dev)
        psum = torch.empty((lib.cream_stem_bwd_chunks(B), N, E), dtype=torch.float32, device=dev)
        _lib.check(lib.cream_stem_bwd(_p(dy), _p(psum), _p(dx0), B, N, E, _stream()), "cream_stem_bwd")
        parts, bparts = linear_wgrad_parts(dy, patches, want_bias=True)
        wshape, bshape, cshape, pshape = ctx.shapes
        if NATIVE_ENDS_GRADS:
            # every parameter gradient of the stem in ONE launch, added into the active slices of the .grad tensors
            pe, (w, b, cls, pos) = ctx.pe, ctx.params
            jobs = GradJobs()
            jobs.add(w, parts, parts.shape[0], E * K, E, K, ld=K)
            jobs.add(b, bparts, bparts.shape[0], E, 1, E)
            chunks = psum.shape[0]
            jobs.add(cls, psum, chunks, N * E, 1, E, ld=cshape[-1])                       # token 0 = class token
            params = [w, b, cls]
            if pos is not None:
                jobs.add(pos, psum, chunks, N * E, N, E, ld=pshape[-1])
                params.append(pos)
            _accumulate_and_notify(jobs, params)
            return None, None, None, None, None, None, None
        gw = torch.zeros((wshape[0], K), dtype=torch.float32, device=dev)
        gb = torch.zeros(bshape, dtype=torch.float32, device=dev)
        torch.sum(parts, dim=0, out=gw[:E])
        torch.sum(bparts, dim=0, out=gb[:E])
        tok = psum.sum(0)                                            # (N, E): sum of dx0 over the batch
        gcls = torch.zeros(cshape, dtype=torch.float32, device=dev)
        gcls.view(-1)[:E] = tok[0]
        gpos = None
        if pshape is not None:
            gpos = torch.zeros(pshape, dtype=torch.float32, device=dev)
            gpos.view(N, pshape[-1])[:, :E] = tok
        return None, gw.view(wshape), gb, gcls, gpos, None, None


def stem_supported(model, x):
    pe = model.patch_embed_super
    E = model.sample_embed_dim[0]
    pos = model.pos_embed if model.abs_pos else None
    if not (patch_embed_supported(pe, x) and x.dtype == torch.float32 and pe.patch_size[1] % 8 == 0 and E % 4 == 0):
        return False
    if pos is not None and (pos.shape[0] != 1 or pos.shape[-1] % 4 or pos.dtype != torch.float32):
        return False
    # PatchembedSuper.forward's own assertion (embedding_super.py:28-31) and the geometry the kernels index with raw
    # pointers: anything else takes the composed path, which asserts
    H, W = x.shape[-2:]
    ph, pw = pe.patch_size
    if (H, W) != tuple(pe.img_size) or H % ph or W % pw:
        return False
    if pos is not None and pos.shape[1] != (H // ph) * (W // pw) + 1:
        return False
    if torch.is_grad_enabled():                     # (see _no_frozen_parameters)
        ends = [pe.proj.weight, pe.proj.bias, model.cls_token] + ([pos] if pos is not None else [])
        if not all(p.requires_grad for p in ends):
            return False
    return not (model.training and (model.sample_dropout or 0.0) > 0.0)


def stem(model, x):
    pe = model.patch_embed_super
    pos = model.pos_embed if model.abs_pos else None
    return StemFunction.apply(x, pe.proj.weight, pe.proj.bias, model.cls_token, pos, pe, model.sample_embed_dim[0])


def patch_embed(pe, x):
    """PatchembedSuper.forward on the device under bf16 autocast (embedding_super.py:27-40)."""
    B, C, H, W = x.shape
    ph, pw = pe.patch_size
    gh, gw = H // ph, W // pw
    E = pe.sample_embed_dim
    patches = x.reshape(B, C, gh, ph, gw, pw).permute(0, 2, 4, 1, 3, 5).reshape(B * gh * gw, C * ph * pw)
    patches = patches.to(torch.bfloat16)
    y = PatchEmbedFunction.apply(patches, pe.proj.weight, pe.proj.bias, pe, E)
    return y.view(B, gh * gw, E)


def patch_embed_supported(pe, x):
    return (x.is_cuda and torch.is_autocast_enabled('cuda') and torch.get_autocast_dtype('cuda') == torch.bfloat16
            and not pe.scale and pe.sample_embed_dim % 8 == 0 and pe.proj.bias is not None
            and (pe.proj.weight.numel() // pe.proj.weight.shape[0]) % 64 == 0)


def operands(blk, fresh=True):
    """The block's BlockOperands (created on first use, re-created when the module moved), refreshed
    if a parameter's version counter moved since the copies were written.  The native optimizer
    (engine.NativeAdamW) rewrites the copies in its own kernel and marks them fresh."""
    ops = blk.__dict__.get(_OPS_KEY)
    if ops is None or ops.key != ops._key():
        ops = blk.__dict__[_OPS_KEY] = BlockOperands(blk)
    if fresh and ops.stale():
        ops.refresh()
    return ops


def refresh_operands(model, force=True):
    """Re-derive the operand copies of every block that has them (checkpoint load, foreign optimizer)."""
    for m in model.modules():
        ops = m.__dict__.get(_OPS_KEY)
        if ops is not None and (force or ops.stale()):
            ops.refresh()


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
            and blk.sample_embed_dim % 8 == 0
            # a hidden width that is not a multiple of 8 (supernet-T: 216 x 3.5 = 756) runs padded, native driver only
            and (blk.sample_ffn_embed_dim_this_layer % 8 == 0
                 or (NATIVE_BLOCK and (blk.sample_ffn_embed_dim_this_layer + 7) // 8 * 8 <= blk.fc1.weight.shape[0]))
            and blk.sample_out_dim == blk.sample_embed_dim
            and a.qkv.bias is not None and a.proj.bias is not None and blk.fc1.bias is not None
            and blk.fc2.bias is not None and _no_frozen_parameters(blk))


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
    ops = operands(blk)
    (wqkv, wproj, w1, w2), (bqkv, bproj, b1, b2) = ops.w, ops.b
    ln1, ln2 = blk.attn_layer_norm, blk.ffn_layer_norm

    if pend is None:
        x = x2d
        a, mean1, rstd1 = ln_fwd(x, ln1.weight[:E], ln1.bias[:E], ln1.eps)
    else:
        x, a, mean1, rstd1 = add_ln_fwd(x2d, pend[0], pend[1], N, ln1.weight[:E], ln1.bias[:E], ln1.eps)
    # qkv rows [q | k | v] = the first Q rows of the de-interleaved parts of the super weight
    # (qkv_super.py:72-77); bias is the plain prefix (qkv_super.py:80-83)
    qkv = linear_fwd_seg(a, wqkv, bqkv, 3 * Q, E, Q)
    tabs = tuple(t.detach() for t in _tables(at))
    o, lse, sp = fused_attention.attn_fwd_raw(qkv.view(B, N, 3, H, 64), *tabs, at.sample_scale, mr, timg=ops.timg)
    p = linear_fwd(o.view(M, Q), wproj, bproj, E, Q)
    x1, c, mean2, rstd2 = add_ln_fwd(x, p, dp1, N, ln2.weight[:E], ln2.bias[:E], ln2.eps)
    h, g = linear_gelu_fwd(c, w1, b1, F_, E)                 # h = gelu'(pre-activation), g = gelu(pre-activation)
    f = linear_fwd(g, w2, b2, E, F_)
    dims = (B, N, E, H, Q, F_, mr, float(at.sample_scale))
    return x1, f, dims, (x, mean1, rstd1, a, qkv, o, lse, sp, x1, mean2, rstd2, c, h, g)


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
    x, mean1, rstd1, a, qkv, o, lse, sp, x1, mean2, rstd2, c, h, g = saved
    wqkv_t, wproj_t, w1_t, w2_t = operands(blk).wt
    ln1, ln2 = blk.attn_layer_norm, blk.ffn_layer_norm

    jobs = GradJobs()
    # ---- MLP branch -----------------------------------------------------------------------
    # every weight gradient as soon as its operands exist, on the side stream (token-sliced partial tiles, added by the
    # finalisation at the end of the block)
    extra = []
    pw2, _ = wgrad_parts_async(df, g)
    jobs.add(blk.fc2.weight, pw2, pw2.shape[0], E * F_, E, F_)
    extra.append(pw2)
    jobs.add(blk.fc2.bias, pb2[0], pb2[1], pb2[2], 1, E, src_offset=pb2[3])
    dh, pb1 = linear_dgrad_mul(df, w2_t, h, E, F_)           # (df . W2) * gelu'(h) (h holds the saved derivative) + fc1 bias partials
    pw1, _ = wgrad_parts_async(dh, c)
    jobs.add(blk.fc1.weight, pw1, pw1.shape[0], F_ * E, F_, E)
    extra.append(pw1)
    jobs.add(blk.fc1.bias, pb1, pb1.shape[0], F_, 1, F_)
    dc = linear_dgrad(dh, w1_t, F_, E)
    # dx1 = dx2 + dLN2(dc); dp = s1 * dx1 is the gradient of the proj output, and its column sums
    # (proj bias) come out of the same pass
    dx1, dp, pl2 = ln_bwd_raw(dc, x1, mean2, rstd2, ln2.weight[:E], dx2, dp1, N, True)
    P = pl2.shape[0]
    jobs.add(ln2.weight, pl2, P, 3 * E, 1, E)
    jobs.add(ln2.bias, pl2, P, 3 * E, 1, E, src_offset=E)
    jobs.add(at.proj.bias, pl2, P, 3 * E, 1, E, src_offset=2 * E)

    # ---- attention branch ---------------------------------------------------------------------
    pwp, _ = wgrad_parts_async(dp, o.view(M, Q))
    jobs.add(at.proj.weight, pwp, pwp.shape[0], E * Q, E, Q)
    extra.append(pwp)
    do = linear_dgrad(dp, wproj_t, E, Q)
    tabs_p = _tables(at)
    dqkv, dtab = fused_attention.attn_bwd_raw(do.view(B, N, H, 64), qkv.view(B, N, 3, H, 64),
                                              *(t.detach() for t in tabs_p), o, lse, sp, scale, mr,
                                              reduce_tables=False, timg=operands(blk).timg)
    nb = tabs_p[0].shape[0]
    for i, t in enumerate(tabs_p):                                     # dtab (workgroup partials, 4, 32, 64)
        jobs.add(t, dtab, dtab.shape[0], 4 * 32 * 64, nb, 64, src_offset=i * 32 * 64)
    dqkv2d = dqkv.view(M, 3 * Q)
    pwq, pbq = wgrad_parts_async(dqkv2d, a, want_bias=True)                  # rows [q | k | v]; bias rides along
    jobs.add(at.qkv.weight, pwq, pwq.shape[0], 3 * Q * E, 3 * Q, E, interleave=Q)
    jobs.add(at.qkv.bias, pbq, pbq.shape[0], 3 * Q, 1, 3 * Q)
    extra += [pwq, pbq]
    da = linear_dgrad_seg(dqkv2d, wqkv_t, 3 * Q, E, Q)
    dx, df_prev, pl1 = ln_bwd_raw(da, x, mean1, rstd1, ln1.weight[:E], dx1, prev_scale, N, want_prev)
    jobs.add(ln1.weight, pl1, P, 3 * E, 1, E)
    jobs.add(ln1.bias, pl1, P, 3 * E, 1, E, src_offset=E)
    finalize_on_side_stream(jobs, blk, [df, g, pb2[0], dh, c, pb1, pl2, dp, o, dtab, dqkv, a, pl1] + extra)
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
    ops = operands(blk)                                              # (refreshes stale copies)
    ent = blk.__dict__.get(_DESC_KEY)
    ln1, ln2 = blk.attn_layer_norm, blk.ffn_layer_norm
    key = (ops.key, ops.w[0].data_ptr(), ln1.weight.data_ptr())      # moves when the module changes device / storage
    if ent is None or ent[1] != key:
        t = _lib.BlockDesc()
        (wq, wp, w1, w2), (wqt, wpt, w1t, w2t), (bq, bp, b1, b2) = ops.w, ops.wt, ops.b
        t.wqkv, t.wqkv_t, t.bqkv = wq.data_ptr(), wqt.data_ptr(), bq.data_ptr()
        t.ld_qkv, t.ld_qkv_t, t.seg_qkv, t.seg_qkv_t = wq.stride(1), wqt.stride(1), wq.stride(0), wqt.stride(0)
        t.wproj, t.wproj_t, t.bproj, t.ld_proj, t.ld_proj_t = wp.data_ptr(), wpt.data_ptr(), bp.data_ptr(), wp.stride(0), wpt.stride(0)
        t.w1, t.w1_t, t.b1, t.ld_w1, t.ld_w1_t = w1.data_ptr(), w1t.data_ptr(), b1.data_ptr(), w1.stride(0), w1t.stride(0)
        t.w2, t.w2_t, t.b2, t.ld_w2, t.ld_w2_t = w2.data_ptr(), w2t.data_ptr(), b2.data_ptr(), w2.stride(0), w2t.stride(0)
        t.ln1_g, t.ln1_b, t.ln2_g, t.ln2_b = (ln1.weight.data_ptr(), ln1.bias.data_ptr(), ln2.weight.data_ptr(),
                                              ln2.bias.data_ptr())
        tabs = _tables(at)
        t.tkv, t.tkh, t.tvv, t.tvh = (x.data_ptr() for x in tabs)
        t.ldt = tabs[0].stride(0)
        t.timg = ops.timg.data_ptr() if ops.timg is not None else 0
        t.eps1, t.eps2 = ln1.eps, ln2.eps
        t.mr = at.max_relative_position
        ent = blk.__dict__[_DESC_KEY] = (t, key)
    d = _lib.BlockDesc.from_buffer_copy(ent[0])
    F = blk.sample_ffn_embed_dim_this_layer
    d.B, d.N, d.E, d.H, d.F = B, N, blk.sample_embed_dim, at.sample_num_heads, (F + 7) // 8 * 8
    d.F_valid = F if F % 8 else 0
    d.gh, d.gw = fused_attention.grid_of(N, d.mr)
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
    lib = _lib.load()
    # (the split counts of the weight gradients are part of the backward layout: cream_gemm_tn8 moves the epoch)
    key = (d.B, d.N, d.E, d.H, d.F, lib.cream_block_layout_epoch())
    hit = _ws_cache.get(key)
    if hit is None:
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
    def forward(ctx, x, scales, blks, *tail_args):
        """With tail_w / tail_b (the final LayerNorm's SUPER parameters) the node also covers the end of
        forward_features (supernet_transformer.py:166-170): returns mean over tokens 1.. of LayerNorm(x) as
        (B, E) fp32 — the last block's output is consumed pending, never materialised (native driver only)."""
        tail_w, tail_b, tail_eps = (tuple(tail_args) + (None, None, 1e-5)[len(tail_args):])[:3]
        ctx.ntail = len(tail_args)
        ctx.tail = tail_w is not None
        if NATIVE_BLOCK:
            return StackFunction._forward_native(ctx, x, scales, blks, tail_w, tail_b, tail_eps)
        assert tail_w is None, "the fused tail needs the native block driver"
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
    def _forward_native(ctx, x, scales, blks, tail_w=None, tail_b=None, tail_eps=1e-5):
        B, N, E = x.shape
        lib = _lib.load()
        x = x.contiguous()
        dev = x.device
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        sc_ptr = scales.data_ptr() if scales is not None else 0
        cur, pend_f, pend_s = x.data_ptr(), 0, 0
        descs, wss, xptrs = [], [], []
        inference = not any(ctx.needs_input_grad)          # (no_grad evaluation: nothing only the backward reads is written)
        for i, blk in enumerate(blks):
            d = _block_desc(blk, B, N)
            d.inference = 1 if inference else 0
            ft, off_x, off_x1, off_f = _ws_layout(d)[:4]
            ws = torch.empty(ft, dtype=torch.uint8, device=dev)
            wp = ws.data_ptr()
            dp1 = sc_ptr + (2 * i) * B * 4 if sc_ptr else 0
            _lib.check(lib.cream_block_fwd(ctypes.byref(d), wp, cur, pend_f, pend_s, dp1, stream), "cream_block_fwd")
            xptrs.append(wp + off_x if pend_f else cur)
            descs.append(d)
            wss.append(ws)
            cur, pend_f, pend_s = wp + off_x1, wp + off_f, (sc_ptr + (2 * i + 1) * B * 4 if sc_ptr else 0)
        ctx.native = True
        ctx.blks = list(blks)
        ctx.descs = descs
        ctx.xptrs = xptrs
        ctx.shape = (B, N, E)
        ctx.has_scales = scales is not None
        if tail_w is not None:
            f32 = dict(dtype=torch.float32, device=dev)
            pooled, xm = torch.empty((B, E), **f32), torch.empty((B, E), **f32)
            stats = torch.empty((2, B * N), **f32)
            part = torch.empty((B, lib.cream_tail_chunks(N), E), **f32)
            gamma, beta = tail_w.detach()[:E].contiguous(), tail_b.detach()[:E].contiguous()
            _lib.check(lib.cream_tail_fwd(pooled.data_ptr(), xm.data_ptr(), part.data_ptr(), stats[0].data_ptr(),
                                          stats[1].data_ptr(), cur, pend_f, pend_s, gamma.data_ptr(), beta.data_ptr(), B, N, E,
                                          tail_eps, stream), "cream_tail_fwd")
            ctx.tail_ptrs = (cur, pend_f, pend_s)
            ctx.tail_shapes = (tuple(tail_w.shape), tuple(tail_b.shape))
            ctx.tail_params = (tail_w, tail_b) if isinstance(tail_w, torch.nn.Parameter) else None
            ctx.save_for_backward(x, *wss, *([scales] if scales is not None else []), xm, stats, gamma)
            return pooled
        out = torch.empty((B, N, E), dtype=torch.float32, device=dev)
        _lib.check(lib.cream_residual_add(out.data_ptr(), cur, pend_f, pend_s, B * N * E, N * E, stream), "cream_residual_add")
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
        join_side_stream_at_end_of_backward(dx.device)   # every parameter gradient of the run is complete when the pass ends
        return (dx.view(B, N, E), None, None) + (None, None, None)[:ctx.ntail]

    @staticmethod
    def _backward_native(ctx, dout):
        blks = ctx.blks
        tens = ctx.saved_tensors
        tail = None
        if ctx.tail:
            tens, tail = tens[:-3], tens[-3:]
        scales = tens[-1] if ctx.has_scales else None
        wss = tens[1:1 + len(blks)]
        B, N, E = ctx.shape
        M = B * N
        lib = _lib.load()
        dev = dout.device
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        if WGRAD_SIDE_STREAM:
            side_st = _side_stream(dev)
            side = ctypes.c_void_p(side_st.cuda_stream)
        else:
            side_st, side = None, stream
        sc_ptr = scales.data_ptr() if scales is not None else 0
        L = len(blks)
        tail_grads = (None, None)
        if tail is not None:
            xm, stats, gamma = tail
            g = dout.contiguous().float()
            dx_t = torch.empty((M, E), dtype=torch.float32, device=dev)
            df_t = torch.empty((M, E), dtype=torch.bfloat16, device=dev)
            part = torch.empty((lib.cream_ln_partials(), E), dtype=torch.float32, device=dev)
            x1p, fp, sp = ctx.tail_ptrs
            _lib.check(lib.cream_tail_bwd(dx_t.data_ptr(), df_t.data_ptr(), part.data_ptr(), g.data_ptr(), x1p, fp,
                                          stats[0].data_ptr(), stats[1].data_ptr(), gamma.data_ptr(), sp, B, N, E, stream),
                       "cream_tail_bwd")
            wshape, bshape = ctx.tail_shapes
            if NATIVE_ENDS_GRADS and ctx.tail_params is not None:
                # dgamma = sum_b g * xm, dbeta = sum_b g: the batch is the "parts" axis of one finalize launch
                tw, tb = ctx.tail_params
                gx = g * xm
                jobs = GradJobs()
                jobs.add(tw, gx, B, E, 1, E)
                jobs.add(tb, g, B, E, 1, E)
                _accumulate_and_notify(jobs, [tw, tb])
            else:
                gw = torch.zeros(wshape, dtype=torch.float32, device=dev)
                gb = torch.zeros(bshape, dtype=torch.float32, device=dev)
                torch.sum(g * xm, dim=0, out=gw[:E])
                torch.sum(g, dim=0, out=gb[:E])
                tail_grads = (gw, gb)
        else:
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
        # (the side stream reads the backward workspaces AND the saved forward workspaces; both are released when this returns)
        join_side_stream_at_end_of_backward(dev, keep + [t for t in tens if t is not None and t.is_cuda])
        out = ws[off_dx:off_dx + M * E * 4].view(torch.float32).view(B, N, E)
        return (out, None, None) + (tail_grads[0], tail_grads[1], None)[:ctx.ntail]


class BlockFunction:
    """One block = a stack of one.  apply(x, dp1, dp2, blk); dp1 / dp2: per-sample drop-path
    scales (B,) or None."""

    @staticmethod
    def apply(x, dp1, dp2, blk):
        scales = None
        if dp1 is not None or dp2 is not None:
            ones = torch.ones(x.shape[0], device=x.device, dtype=torch.float32)
            scales = torch.stack([dp1 if dp1 is not None else ones, dp2 if dp2 is not None else ones]).unsqueeze(0)
        return StackFunction.apply(x, scales, [blk], None, None, 1e-5)

"""Affinity-mimicking distillation loss of TinyCLIP — host-side mirror of
TinyCLIP/src/open_clip/clip_soft_loss.py:10-88 (`ClipSoftLoss`) and of the feature gathering it
calls (TinyCLIP/src/open_clip/loss.py:71-106, `gather_feature`).

Per step and rank (local_loss=True, the only mode the reference's class accepts, :29):
    logits_i2t = s * I_local @ T_all^T        logits_t2i = s * T_local @ I_all^T        (student)
    the same with the frozen teacher's features and logit scale
    loss = ( CE(logits_i2t, softmax(teacher_i2t)) + CE(logits_t2i, softmax(teacher_t2i)) ) / 2
where *_all are the features of ALL ranks (global batch) — the one exchange step of this path.

MI355X / RCCL shape of the exchange (not a translation of the reference's four list-based
all_gathers, loss.py:94-104): image and text features of a tower pair travel TOGETHER — one
`all_gather_into_tensor` of a (B_local, 2 D) buffer for the student (differentiable: its backward is
one reduce-scatter of the gathered gradient, i.e. what `torch.distributed.nn.all_gather` does in
world many sends) and one for the teacher (no gradient) — 2 collectives per step instead of 4, each
a single contiguous message over the xGMI mesh.
"""
import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F


class _AllGatherCat(torch.autograd.Function):
    """(B, D) on every rank -> (world * B, D), rank-major.  Backward: the gradient of the gathered
    tensor is summed over ranks and each rank keeps its own rows (reduce-scatter)."""

    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        world = dist.get_world_size(group)
        out = x.new_empty((world * x.shape[0],) + tuple(x.shape[1:]))
        dist.all_gather_into_tensor(out, x.contiguous(), group=group)
        return out

    @staticmethod
    def backward(ctx, g):
        group = ctx.group
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        g = g.contiguous()
        b = g.shape[0] // world
        if dist.get_backend(group) == "nccl":
            mine = g.new_empty((b,) + tuple(g.shape[1:]))
            dist.reduce_scatter_tensor(mine, g, op=dist.ReduceOp.SUM, group=group)
            return mine, None
        dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group)        # backends without reduce_scatter (gloo)
        return g[rank * b:(rank + 1) * b].clone(), None


class _GatherStarted(torch.autograd.Function):
    """The differentiable view of a gather that was posted asynchronously and has COMPLETED: forward hands out the receive buffer,
    backward is _AllGatherCat's reduce-scatter."""

    @staticmethod
    def forward(ctx, x, out, group):
        ctx.group = group
        return out.view_as(out)

    @staticmethod
    def backward(ctx, g):
        return _AllGatherCat.backward(ctx, g)[0], None, None


class FeatureGather:
    """Both towers' features of every rank with ONE collective that runs UNDER the local block of the similarity products
    (all-gather / similarity-GEMM overlap): start() posts the asynchronous all-gather of the (B_local, 2 D) buffer —
    RCCL runs it on its own stream over xGMI —, the caller multiplies its LOCAL rows against its LOCAL columns meanwhile (they need
    nothing from the other ranks), wait() orders the current stream behind the collective and returns (all_image, all_text)."""

    def __init__(self, image_features, text_features, with_grad=True, group=None):
        self.d = image_features.shape[1]
        self.local = (image_features, text_features)
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        self.work = None
        if self.world == 1:
            return
        both = torch.cat([image_features, text_features], dim=1).contiguous()
        out = both.new_empty((self.world * both.shape[0], both.shape[1]))
        self.work = dist.all_gather_into_tensor(out, both.detach(), group=group, async_op=True)
        self._both, self._out, self._group = both, out, group      # (the send buffer must outlive the collective)
        self._with_grad = with_grad and both.requires_grad

    def wait(self):
        if self.world == 1:
            return self.local
        self.work.wait()
        # the differentiable view is attached once the buffer is complete (autograd forbids in-place writes under a custom view)
        allf = _GatherStarted.apply(self._both, self._out, self._group) if self._with_grad else self._out
        return allf[:, :self.d], allf[:, self.d:]


def overlapped_similarities(image_features, text_features, with_grad=True, group=None):
    """(image_local . all_text^T, text_local . all_image^T) with the feature gather running under the two LOCAL blocks.  Column
    blocks of an NT product are independent (same contraction order per element), so the values equal the unoverlapped path's."""
    g = FeatureGather(image_features, text_features, with_grad, group)
    if g.world == 1:
        return similarity(image_features, text_features), similarity(text_features, image_features)
    # rows x OWN columns: needs nothing from the other ranks.  Without gather_with_grad the column operand carries no gradient, the
    # local block included (loss.py:96-103 puts the local tensor back only when `not local_loss`; ClipSoftLoss asserts local_loss)
    col_t = text_features if with_grad else text_features.detach()
    col_i = image_features if with_grad else image_features.detach()
    loc_it = similarity(image_features, col_t)
    loc_ti = similarity(text_features, col_i)
    all_image, all_text = g.wait()
    b, r, w = image_features.shape[0], g.rank, g.world

    def assemble(a, all_b, local_block):
        parts = []
        if r > 0:
            parts.append(similarity(a, all_b[:r * b]))
        parts.append(local_block)
        if r < w - 1:
            parts.append(similarity(a, all_b[(r + 1) * b:]))
        return torch.cat(parts, dim=1)

    return assemble(image_features, all_text, loc_it), assemble(text_features, all_image, loc_ti)


def gather_features_with_grad(image_features, text_features, with_grad=True, group=None):
    """-> (all_image, all_text): both towers' features of every rank with ONE collective."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return image_features, text_features
    d = image_features.shape[1]
    both = torch.cat([image_features, text_features], dim=1)
    if with_grad:
        allf = _AllGatherCat.apply(both, group)
    else:
        with torch.no_grad():
            allf = _AllGatherCat.apply(both.detach(), group)
    return allf[:, :d], allf[:, d:]


class _SimilarityNT(torch.autograd.Function):
    """S (M, N) = A (M, D) . B (N, D)^T in bf16 on the framework's OWN matrix-core GEMMs (csrc/gemm_mfma.hpp) — the four
    similarity products of ClipSoftLoss (clip_soft_loss.py:34-52), forward and both gradients (dA = dS . B through the NT
    kernel on B^T, dB = dS^T . A through the split-K TN kernel): no vendor GEMM library on this path either."""

    @staticmethod
    def forward(ctx, a, b):
        from ..autoformer import block as K
        a16, b16 = a.to(torch.bfloat16).contiguous(), b.to(torch.bfloat16).contiguous()
        ctx.save_for_backward(a16, b16)
        ctx.dtypes = (a.dtype, b.dtype)
        with torch.cuda.device(a.device):
            return K.linear_fwd(a16, b16, None, b16.shape[0], a16.shape[1])

    @staticmethod
    def backward(ctx, ds):
        from ..autoformer import block as K
        a16, b16 = ctx.saved_tensors
        ds = ds.to(torch.bfloat16).contiguous()
        da = db = None
        with torch.cuda.device(ds.device):
            if ctx.needs_input_grad[0]:
                da = K.linear_dgrad(ds, b16.t().contiguous(), b16.shape[0], a16.shape[1]).to(ctx.dtypes[0])
            if ctx.needs_input_grad[1]:
                db = K.wgrad(ds, a16).to(ctx.dtypes[1])
        return da, db


def similarity(a, b):
    """a (M, D) . b (N, D)^T.  Device tensors go through the own kernels: bf16 under autocast (what `@` would compute in),
    exact fp32 otherwise (csrc/gemm_f32.hip); host tensors (tests, the gloo runs) keep the framework product."""
    if a.is_cuda and a.shape[1] % 8 == 0 and b.shape[0] % 8 == 0:
        bf16_autocast = torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16
        if bf16_autocast or a.dtype == torch.bfloat16:
            return _SimilarityNT.apply(a, b)
        if torch.is_autocast_enabled("cuda"):                 # fp16 autocast: what `@` computes in — not this library's path
            return a @ b.T
        from ..autoformer import native_fp32
        if native_fp32.usable(a, b):
            return native_fp32.linear(a, b, None, b.shape[0], a.shape[1])
    return a @ b.T


class ClipSoftLoss(nn.Module):
    """Same constructor flags and forward signature as the reference's ClipSoftLoss
    (clip_soft_loss.py:11-32, :54-88); `use_horovod` is accepted and must be False (RCCL only)."""

    def __init__(self, local_loss=True, gather_with_grad=False, cache_labels=False, rank=None, world_size=None,
                 use_horovod=False, group=None):
        super().__init__()
        assert local_loss, "the reference's ClipSoftLoss only supports local_loss (clip_soft_loss.py:29)"
        assert not use_horovod, "RCCL through torch.distributed only"
        self.local_loss, self.gather_with_grad = local_loss, gather_with_grad
        self.group = group
        self.overlap = True                   # the feature gather runs under the local similarity blocks (False: gather, then multiply)

    def compute_sim(self, image_features, text_features, with_grad):
        if self.overlap:
            return overlapped_similarities(image_features, text_features, with_grad, self.group)
        all_image, all_text = gather_features_with_grad(image_features, text_features, with_grad, self.group)
        return similarity(image_features, all_text), similarity(text_features, all_image)

    def forward(self, image_features, text_features, logit_scale, teacher_image_features, teacher_text_features,
                teacher_logit_scale, average_two_losses=True, labels=None):
        li, lt = self.compute_sim(image_features, text_features, self.gather_with_grad)
        ti, tt = self.compute_sim(teacher_image_features, teacher_text_features, False)
        li, lt = logit_scale * li, logit_scale * lt
        ti, tt = teacher_logit_scale * ti, teacher_logit_scale * tt

        def single(logits, teacher_logits):
            return F.cross_entropy(logits, F.softmax(teacher_logits, -1))

        if average_two_losses:
            return (single(li, ti) + single(lt, tt)) / 2
        return single(li, ti), single(lt, tt)

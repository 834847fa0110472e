"""Fused iRPE attention (contextual rpe on q / k / v) — host side of csrc/irpe_attn.hip.

`attention(qkv, scale, rpe_q, rpe_k, rpe_v)` computes what `RPEAttention.forward` does between the
qkv and proj linears (iRPE/DeiT-with-iRPE/rpe_vision_transformer.py:68-97) in one launch forward and
two launches (+ one tiny table-gradient product per rpe) backward; nothing of size L^2 is written to
HBM.  It takes the (B, L, 3, H, 64) bf16 output of the qkv linear as it is (q, k, v are strided views)
and returns (B, L, H*64) ready for the proj linear.

`usable(...)` says whether a given RPEAttention configuration is covered: bf16 operands (autocast),
head_dim 64, each rpe either absent or an iRPE with at most 64 buckets (product: 50, euclidean /
quant: <= 64; cross: rows + cols as ONE table over the occurring bucket pairs, 50 in the zoo — iRPE_Cross.merged_table) —
contextual, or bias mode on q / k (the lookups are then the bias table itself, irpe.py:622-624).  Everything else (fp32) takes
the composed path of cream_amd.rpe_attention on the HIP rpe_index operator.
"""
import ctypes
import os

import torch

from . import _lib, timing

_BYTES = {}          # (data_ptr, shape, device) -> (ids tensor kept alive, as-is uint8, transposed uint8)


def padded_len(L):
    return (L + 31) // 32 * 32


def bucket_bytes(ids):
    """int32 (L, L) bucket ids -> (NP x NP uint8 as is, NP x NP uint8 transposed), cached per table."""
    key = (ids.data_ptr(), tuple(ids.shape), str(ids.device))
    hit = _BYTES.get(key)
    if hit is not None:
        return hit[1], hit[2]
    assert ids.dtype == torch.int32 and ids.dim() == 2 and ids.is_contiguous() and ids.is_cuda
    Lq, Lk = ids.shape
    NP = padded_len(max(Lq, Lk))
    lib = _lib.load()
    outs = []
    with torch.cuda.device(ids.device):
        stream = torch.cuda.current_stream().cuda_stream
        for tr in (0, 1):
            dst = torch.empty((NP, NP), dtype=torch.uint8, device=ids.device)
            _lib.check(lib.cream_irpe_bucket_bytes(dst.data_ptr(), ids.data_ptr(), Lq, Lk, NP, tr, stream),
                       "cream_irpe_bucket_bytes")
            outs.append(dst)
    _BYTES[key] = (ids, outs[0], outs[1])
    return outs[0], outs[1]


def _term(rpe, L, device):
    """-> (table parameter, head stride, query-major ids, key-major ids, nb, bias mode) of one rpe module."""
    if rpe is None:
        return None
    from .irpe import iRPE_Cross
    if type(rpe) is iRPE_Cross:          # rows + cols = one lookup over the occurring bucket pairs (irpe.py:758-760)
        ids, _, _, nb = rpe.merged_ids_for(L, device)
        asis, tr = bucket_bytes(ids)
        w = rpe.merged_table(L, device)  # differentiable: the table gradient reaches both parameters through it
        return w, (0 if w.shape[0] == 1 else w[0].numel()), asis, tr, nb, rpe.mode == "bias"
    ids = rpe.bucket_ids_for(L, device)
    asis, tr = bucket_bytes(ids)
    bias = rpe.mode == "bias"
    w = rpe.lookup_table_bias if bias else rpe.lookup_table_weight
    hs = 0 if w.shape[0] == 1 else w[0].numel()
    return w, hs, asis, tr, rpe.num_buckets, bias


def usable(qkv_dtype, device, head_dim, L, rpes, attn_drop_active=False, dropout_p=0.0):
    """(attention dropout no longer excludes the fused path: the kernels regenerate the keep mask from a seed;
    `nn.Dropout(p=1.0)` is legal and stays on the composed path — the C ABI rejects dropout_p >= 1)"""
    if os.environ.get("CREAM_IRPE_FUSED", "1") == "0":
        return False
    if dropout_p >= 1.0:
        return False
    if device.type != "cuda" or qkv_dtype != torch.bfloat16 or head_dim != 64 or L > 2048:
        return False
    from .irpe import iRPE, iRPE_Cross
    nbs = set()
    for r in rpes:
        if r is None:
            continue
        if type(r) is iRPE_Cross:
            parts, nb = (r.rp_rows, r.rp_cols), r.merged_ids_for(L, device)[3]
        else:
            parts, nb = (r,), r.num_buckets
        for m in parts:
            if type(m) is not iRPE or m.mode not in ("contextual", "bias"):
                return False
            w = m.lookup_table_bias if m.mode == "bias" else m.lookup_table_weight
            if w.dtype != torch.float32 or not w.is_contiguous():
                return False
        if nb > 64:
            return False
        nbs.add(nb)
    return len(nbs) <= 1


def _desc(qkv, scale, terms, out, lse, sv):
    B, L, _, H, D = qkv.shape
    d = _lib.IrpeAttnDesc()
    es = qkv.element_size()
    base = qkv.data_ptr()
    sb, sn, s3, sh, _ = qkv.stride()
    d.q, d.k, d.v = base, base + s3 * es, base + 2 * s3 * es
    d.sb, d.sn, d.sh = sb, sn, sh
    d.out, d.lse, d.sv = out.data_ptr(), lse.data_ptr(), (sv.data_ptr() if sv is not None else None)
    tq, tk, tv = terms
    nb = 1
    if tq is not None:                       # rpe_q: bucket_q[j][i] is key-major as stored
        d.idq, d.idq_t, nb = tq[3].data_ptr(), tq[2].data_ptr(), tq[4]
        if tq[5]:
            d.bq, d.bq_hs = tq[0].data_ptr(), tq[1]
        else:
            d.wq, d.wq_hs = tq[0].data_ptr(), tq[1]
    if tk is not None:
        d.idk, d.idk_t, nb = tk[2].data_ptr(), tk[3].data_ptr(), tk[4]
        if tk[5]:
            d.bk, d.bk_hs = tk[0].data_ptr(), tk[1]
        else:
            d.wk, d.wk_hs = tk[0].data_ptr(), tk[1]
    if tv is not None:
        d.wv, d.wv_hs, d.idv, d.idv_t, nb = tv[0].data_ptr(), tv[1], tv[2].data_ptr(), tv[3].data_ptr(), tv[4]
    d.B, d.H, d.L, d.NP, d.nb = B, H, L, padded_len(L), nb
    d.scale = scale
    return d


def dropout_keep_mask(seed, B, H, L):
    """The keep mask of the kernels' attention dropout as a (B, H, L, L) uint32 array of hash values: element (b, h, i, j)
    is kept iff value >= round(p * 2^32).  numpy restatement of `drop_key` / `drop_keep` in csrc/irpe_attn.hip (a 32-bit
    multiply-xorshift mix of a per-(b, h) key and (i << 16 | j)); used by the tests to build the reference."""
    import numpy as np

    def mix32(x):
        x = x.astype(np.uint64)
        x ^= x >> np.uint64(16); x = (x * np.uint64(0x7feb352d)) & np.uint64(0xFFFFFFFF)
        x ^= x >> np.uint64(15); x = (x * np.uint64(0x846ca68b)) & np.uint64(0xFFFFFFFF)
        x ^= x >> np.uint64(16)
        return x

    bh = np.arange(B * H, dtype=np.uint64)
    key = mix32(np.uint64(seed & 0xFFFFFFFF) ^ (((bh + np.uint64(1)) * np.uint64(0x9E3779B9)) & np.uint64(0xFFFFFFFF)))
    i = np.arange(L, dtype=np.uint64)[:, None]
    j = np.arange(L, dtype=np.uint64)[None, :]
    pair = (i << np.uint64(16)) | j
    return mix32(key[:, None, None] ^ pair[None]).astype(np.uint32).reshape(B, H, L, L)


def dropout_threshold(p):
    """round(p * 2^32) clamped to [1, 2^32 - 1] as the C ABI computes it (cream_irpe_attn_desc.dropout_p is a float)."""
    import numpy as np
    thr = float(np.float32(p)) * 4294967296.0
    return int(min(max(thr, 1.0), 4294967295.0))


_seed_gen = None


def _new_seed(device=None):
    """Seed of one in-kernel keep mask.  On a GPU it is derived from the device generator's (seed, offset) — the state
    torch.nn.Dropout itself consumes (rpe_vision_transformer.py:64, :91) — and advances that offset: masks follow
    torch.manual_seed, a second torch.manual_seed(s) with the SAME s replays them, and no number of the host stream is used
    (Mixup / the configuration sampler draw from it between layers).  Without a device generator: a dedicated host generator
    seeded from torch.initial_seed()."""
    global _seed_gen
    gens = getattr(torch.cuda, "default_generators", ())
    if torch.cuda.is_available() and len(gens):
        g = gens[torch.cuda.current_device() if device is None else torch.device(device).index or 0]
        if hasattr(g, "get_offset"):
            seed, off = int(g.initial_seed()), int(g.get_offset())
            g.set_offset(off + 4)                              # (multiples of 4: one Philox counter step, as a dropout launch takes)
            x = (seed * 0x9E3779B97F4A7C15 + (off + 1) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
            x ^= x >> 31
            return int(x % (2 ** 31 - 1))
    if _seed_gen is None or _seed_gen[0] != torch.initial_seed():
        g = torch.Generator()
        g.manual_seed(torch.initial_seed() ^ 0x5DEECE66D)
        _seed_gen = (torch.initial_seed(), g)
    return int(torch.randint(0, 2 ** 31 - 1, (1,), generator=_seed_gen[1]).item())


def _flops(B, H, L, n_terms, bwd):
    return (2.5 if bwd else 1.0) * 4.0 * B * H * L * L * 64 + (3 if bwd else 1) * n_terms * 2.0 * B * H * L * 64 * 64


def fwd_core(qkv, scale, terms, drop_p=0.0, seed=0, causal=False):
    """One forward launch: qkv (B, L, 3, H, 64) bf16 -> (out (B, L, H*64), lse (B, H, L) fp32, sv (B, H, NP, 64) or None).
    `terms` = (_term(rpe_q), _term(rpe_k), _term(rpe_v)), each None when absent.  Outside autograd."""
    B, L, _, H, D = qkv.shape
    NP = padded_len(L)
    out = torch.empty((B, L, H * D), dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty((B, H, L), dtype=torch.float32, device=qkv.device)
    sv = torch.empty((B, H, NP, 64), dtype=qkv.dtype, device=qkv.device) if terms[2] is not None else None
    d = _desc(qkv, scale, terms, out, lse, sv)
    d.dropout_p, d.dropout_seed = float(drop_p), int(seed)
    d.causal = 1 if causal else 0
    n_terms = sum(t is not None for t in terms)
    with torch.cuda.device(qkv.device), timing.region("irpe_attn_fwd", flops=_flops(B, H, L, n_terms, False)):
        rc = _lib.load().cream_irpe_attn_fwd(ctypes.byref(d), torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "cream_irpe_attn_fwd")
    return out, lse, sv


def bwd_core(dout, qkv, out, lse, sv, scale, terms, drop_p=0.0, seed=0, causal=False):
    """The backward launches of fwd_core: -> (dqkv (B, L, 3, H, 64), [d table of rpe_q, rpe_k, rpe_v] — each None or a tensor of the
    table's shape and dtype).  Outside autograd."""
    tq, tk, tv = terms
    if tv is None:
        sv = None
    B, L, _, H, D = qkv.shape
    NP = padded_len(L)
    dev = qkv.device
    dout = dout.contiguous()
    dqkv = torch.empty_like(qkv, memory_format=torch.contiguous_format)
    d = _desc(qkv, scale, terms, out, lse, sv)
    d.dropout_p, d.dropout_seed = float(drop_p), int(seed)
    d.causal = 1 if causal else 0
    d.dout = dout.data_ptr()
    es = dqkv.element_size()
    sb, sn, s3, sh, _ = dqkv.stride()
    d.dq, d.dk, d.dv = dqkv.data_ptr(), dqkv.data_ptr() + s3 * es, dqkv.data_ptr() + 2 * s3 * es
    d.dsb, d.dsn, d.dsh = sb, sn, sh
    delta = torch.empty((B, H, NP), dtype=torch.float32, device=dev)
    rows = lambda: torch.empty((B, H, NP, 64), dtype=qkv.dtype, device=dev)      # noqa: E731
    lkg = dlk = gg = dlq = None
    d.delta = delta.data_ptr()
    if tk is not None:
        lkg, dlk = rows(), rows()
        d.lkg, d.dlk = lkg.data_ptr(), dlk.data_ptr()
    if tv is not None:
        gg = rows()
        d.gg = gg.data_ptr()
    if tq is not None:
        dlq = rows()
        d.dlq = dlq.data_ptr()
    lib = _lib.load()
    n_terms = sum(t is not None for t in terms)
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream().cuda_stream
        with timing.region("irpe_attn_bwd", flops=_flops(B, H, L, n_terms, True)):
            rc = lib.cream_irpe_attn_bwd(ctypes.byref(d), stream)
        _lib.check(rc, "cream_irpe_attn_bwd")

        def table_grad(x, xs, y, ys, mul):
            part = torch.empty((B, H, 64, 64), dtype=torch.float32, device=dev)
            _lib.check(lib.cream_irpe_table_grad(part.data_ptr(), x, xs[0], xs[1], xs[2], y, ys[0], ys[1], ys[2],
                                                 B, H, L, mul, stream), "cream_irpe_table_grad")
            return part.sum(0)                                   # (H, 64, 64)

        qs = qkv.stride()
        q_str, row_str = (qs[0], qs[1], qs[3]), (H * NP * 64, 64, NP * 64)
        do_str = (L * H * 64, H * 64, 64)
        es_q = qkv.element_size()
        grads = [None, None, None]
        if tq is not None and tq[5]:   # bias mode: d lookup_table_bias (H', nb) = the bucket gradient rows summed
            grads[0] = dlq[:, :, :L].sum((0, 2), dtype=torch.float32)
        elif tq is not None:    # d lookup_table_weight(rpe_q) (H', 64, nb) = (scale k)^T dlq
            grads[0] = table_grad(qkv.data_ptr() + qs[2] * es_q, q_str, dlq.data_ptr(), row_str, scale)
        if tk is not None and tk[5]:
            grads[1] = dlk[:, :, :L].sum((0, 2), dtype=torch.float32)
        elif tk is not None:    # (scale q)^T dlk
            grads[1] = table_grad(qkv.data_ptr(), q_str, dlk.data_ptr(), row_str, scale)
        if tv is not None:      # (H', nb, 64) = sv^T dout
            grads[2] = table_grad(sv.data_ptr(), row_str, dout.data_ptr(), do_str, 1.0)
    res = []
    for gpart, t, transposed in zip(grads, terms, (True, True, False)):
        if gpart is None:
            res.append(None)
            continue
        w, nb = t[0], t[4]
        if w.shape[0] == 1:
            gpart = gpart.sum(0, keepdim=True)
        if t[5]:
            res.append(gpart[:, :nb].to(w.dtype).contiguous())
            continue
        res.append((gpart[:, :, :nb] if transposed else gpart[:, :nb, :]).to(w.dtype).contiguous())
    return dqkv, res


class _Fused(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, scale, wq, wk, wv, terms, drop_p=0.0, seed=0):
        out, lse, sv = fwd_core(qkv, scale, terms, drop_p, seed)
        ctx.drop = (float(drop_p), int(seed))
        ctx.save_for_backward(qkv, out, lse, sv if sv is not None else lse)
        ctx.scale, ctx.terms = scale, terms
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, lse, sv = ctx.saved_tensors
        dqkv, res = bwd_core(dout, qkv, out, lse, sv, ctx.scale, ctx.terms, *ctx.drop)
        return dqkv, None, res[0], res[1], res[2], None, None, None


_NO_TERMS = (None, None, None)


def plain_fwd(qkv, scale, causal=False):
    """Attention without any relative position term, forward only, outside autograd: qkv (B, L, 3, H, 64) bf16 ->
    (out (B, L, H*64) bf16, lse (B, H, L) fp32).  For callers that sequence their own backward (cream_amd.tinyclip.native)."""
    return fwd_core(qkv, scale, _NO_TERMS, causal=causal)[:2]


def plain_bwd(dout, qkv, out, lse, scale, causal=False):
    """Backward of plain_fwd: -> dqkv (B, L, 3, H, 64) bf16 (two launches)."""
    return bwd_core(dout, qkv, out, lse, None, scale, _NO_TERMS, causal=causal)[0]


def attention(qkv, scale, rpe_q, rpe_k, rpe_v, dropout_p=0.0, seed=None):
    """qkv (B, L, 3, H, 64) bf16 -> (B, L, H*64).  dropout_p > 0: attention dropout inside the kernels
    (rpe_vision_transformer.py:86); `seed` (default: drawn from torch's host generator) fixes the keep mask, which
    `dropout_keep_mask` reproduces."""
    assert qkv.dim() == 5 and qkv.shape[2] == 3 and qkv.shape[4] == 64 and qkv.stride(4) == 1
    L, dev = qkv.shape[1], qkv.device
    terms = tuple(_term(r, L, dev) for r in (rpe_q, rpe_k, rpe_v))
    ws = [t[0] if t is not None else None for t in terms]
    if dropout_p and seed is None:
        seed = _new_seed(dev)
    return _Fused.apply(qkv, float(scale), ws[0], ws[1], ws[2], terms, float(dropout_p or 0.0), int(seed or 0))

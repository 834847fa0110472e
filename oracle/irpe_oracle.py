"""TEST INFRASTRUCTURE — CPU (PyTorch fp32 / numpy) restatement of iRPE's product-method bucket table and of one
`RPEAttention` layer in the reference's own PURE-PYTORCH formulation (the path the reference takes when its
`rpe_index_cpp` extension is not built: the flat-index gather of irpe.py:646-647, the dense gathered weight of
irpe.py:683-687).  Deliberately NOT the fused algebra of csrc/irpe_attn.hip.

Pinned by tests/test_irpe_cpu.py::test_irpe_oracle_is_pinned against fixtures that tests/golden/make_golden.py
produced by running the reference's own classes: the bucket tables (`irpe_buckets.*`: sums 941,241 / 8,019,121)
and an RPEAttention forward / backward (`irpe_attention.npz`).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this file.  Paths below are relative to iRPE/DeiT-with-iRPE/.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def piecewise_index(rel, alpha, beta, gamma):
    """irpe.py:18-52 (Eq. 18 of the paper) for integer offsets: identity up to alpha, logarithmic beyond, clipped at beta."""
    rel = np.asarray(rel, dtype=np.int64)
    a = np.abs(rel).astype(np.float32)
    out = rel.copy()
    far = a > alpha
    with np.errstate(divide="ignore"):
        # the reference evaluates this in float32 torch arithmetic on integer inputs (log of int64 -> float32)
        y = torch.from_numpy(a[far])
        y = (alpha + torch.log(y / alpha) / math.log(gamma / alpha) * (beta - alpha)).round().clip(max=beta)
    out[far] = (np.sign(rel[far]) * y.numpy()).astype(np.int64)
    return out


def product_bucket_ids(height, width, skip, ratio=1.9):
    """irpe.py:176-204 (`_rp_2d_product`), :330-361 (offsets of every token pair) and :364-420 (the extra bucket of
    the class token) with the coefficients of get_single_rpe_config (:808-810: alpha, beta, gamma = 1, 2, 8 x ratio).
    -> (ids (skip + h w, skip + h w) int64, num_buckets)."""
    alpha, beta, gamma = 1 * ratio, 2 * ratio, 8 * ratio
    beta_int = int(beta)
    S = 2 * beta_int + 1
    rows = np.repeat(np.arange(height), width)
    cols = np.tile(np.arange(width), height)
    dr = rows[:, None] - rows[None, :]
    dc = cols[:, None] - cols[None, :]
    r = piecewise_index(dr, alpha, beta, gamma) + beta_int
    c = piecewise_index(dc, alpha, beta, gamma) + beta_int
    ids = r * S + c
    nb = S * S
    if skip > 0:
        L = height * width
        full = np.full((skip + L, skip + L), nb, dtype=np.int64)
        full[skip:, skip:] = ids
        ids, nb = full, nb + 1
    return ids, nb


def rpe_attention_layer(p, x, num_heads, ids, nb):
    """rpe_vision_transformer.py:68-97 with contextual product rpe on whichever of q / k / v has a table in `p`
    (`rpe_q.lookup_table_weight` (1 or H, 64, nb), `rpe_k...`, `rpe_v.lookup_table_weight` (1 or H, nb, 64)) in the
    reference's fallback formulation: lookup = x W (irpe.py:641-644), gathered with the flat index
    i * nb + ids[i, j] (irpe.py:573-583, :646-647); value side = the dense gathered weight (irpe.py:683-687)."""
    B, N, C = x.shape
    H, hd = num_heads, C // num_heads
    scale = hd ** -0.5
    qkv = F.linear(x, p["qkv.weight"], p.get("qkv.bias")).reshape(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * scale, qkv[1], qkv[2]
    ids_t = torch.as_tensor(ids, dtype=torch.long)
    flat = (torch.arange(N).view(N, 1) * nb + ids_t).flatten()                   # irpe.py:573-583

    def transposed(xh, w):                                                       # irpe.py:641-647
        lookup = torch.matmul(xh.transpose(0, 1).reshape(-1, B * N, hd), w).view(-1, B, N, nb).transpose(0, 1)
        return lookup.flatten(2)[:, :, flat].view(B, -1, N, N)

    attn = q @ k.transpose(-2, -1)
    if "rpe_k.lookup_table_weight" in p:
        attn = attn + transposed(q, p["rpe_k.lookup_table_weight"])
    if "rpe_q.lookup_table_weight" in p:
        attn = attn + transposed(k * scale, p["rpe_q.lookup_table_weight"]).transpose(2, 3)
    attn = attn.softmax(dim=-1)
    out = attn @ v
    if "rpe_v.lookup_table_weight" in p:
        w = p["rpe_v.lookup_table_weight"]
        weight = w[:, ids_t.flatten()].view(w.shape[0], N, N, hd)                # irpe.py:683-685
        out = out + torch.matmul(attn.permute(1, 2, 0, 3), weight).permute(2, 0, 1, 3)   # :686-687
    return F.linear(out.transpose(1, 2).reshape(B, N, C), p["proj.weight"], p["proj.bias"])

"""Fused iRPE attention (csrc/irpe_attn.hip) on the MI355X against an fp32 restatement of
RPEAttention.forward's core (rpe_vision_transformer.py:68-97 + irpe.py:585-687) evaluated on the SAME
bf16-rounded inputs — forward, dq/dk/dv and the three lookup-table gradients — for every subset of
rpe_q / rpe_k / rpe_v, shared and per-head tables, contextual and bias mode (irpe.py:622-624), L = 50 / 197 / 577.  The pinned comparison with the
reference itself is tests/test_irpe_gpu.py::test_rpe_attention_L577_on_gpu (reference-made fixture; under
autocast RPEAttention takes this kernel)."""
import pytest
import torch


pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def max_rel(a, b):
    """max |a - b| / max |b| (the measure of tests/helpers.py), on the device."""
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def _restatement(qkv, scale, mods, round_lookups=True, keep=None, ids=None):
    """round_lookups: the lookups (s q) W leave the reference's autocast matmul as 16-bit values (irpe.py:646 under amp) and the
    kernel keeps them as bf16 rows in LDS; False evaluates the same algebra in pure fp32 (reported next to the asserted comparison)."""
    rnd = (lambda t: t.to(torch.bfloat16).float()) if round_lookups else (lambda t: t)
    q, k, v = qkv.float().permute(2, 0, 3, 1, 4).unbind(0)                    # (B, H, L, 64)
    rq, rk, rv = mods
    L = q.shape[2]
    qs = q * scale
    a = qs @ k.transpose(-2, -1)

    def ids_of(m):                                  # `ids`: the oracle's bucket table (config 4) instead of the product's
        return ids if ids is not None else m.bucket_ids_for(L, qkv.device).long()

    def w_of(m):
        w = m.lookup_table_weight.float()
        return w[0] if w.shape[0] == 1 else w.unsqueeze(0)

    def bias_of(m):                                                            # (1, H', L, L), irpe.py:622-624
        return m.lookup_table_bias.float()[:, ids_of(m).flatten()].view(1, -1, L, L)

    def parts(m):                                   # iRPE_Cross = rows + cols, two plain iRPE modules (irpe.py:758-760)
        return [] if m is None else ([m.rp_rows, m.rp_cols] if hasattr(m, "rp_rows") else [m])

    for m in parts(rk):
        if m.mode == "bias":
            a = a + bias_of(m)
        else:
            lk = rnd(qs @ w_of(m))                                             # the autocast matmul's output dtype
            a = a + lk.gather(-1, ids_of(m).expand(*lk.shape[:2], L, L))
    for m in parts(rq):
        if m.mode == "bias":
            a = a + bias_of(m).transpose(2, 3)
        else:
            lq = rnd((k * scale) @ w_of(m))
            a = a + lq.gather(-1, ids_of(m).expand(*lq.shape[:2], L, L)).transpose(2, 3)
    p = a.softmax(-1)
    if keep is not None:                                                       # attn_drop (:86): mask / (1 - rate), given
        p = p * keep
    out = p @ v
    for m in parts(rv):
        sv = torch.zeros(*p.shape[:3], m.num_buckets, device=p.device).scatter_add_(-1, ids_of(m).expand_as(p), p)
        out = out + sv @ w_of(m)
    return out.transpose(1, 2).reshape(q.shape[0], L, -1)


CASES = [("k", True, 50, "product"), ("k", False, 197, "product"), ("q", True, 50, "product"), ("v", True, 50, "product"),
         ("qk", True, 197, "product"), ("kv", False, 50, "product"), ("qkv", True, 197, "product"), ("qkv", False, 50, "product"),
         ("k", True, 577, "product"), ("qkv", True, 577, "product"), ("", True, 50, "product"),
         ("qkv", True, 197, "euc"), ("qkv", False, 197, "quant"), ("k", True, 196, "euc"),      # 196: skip = 0 (no class token)
         ("k", True, 197, "product", "bias"), ("qk", False, 197, "product", "bias"), ("q", False, 50, "quant", "bias"),
         ("qkv", False, 577, "product", "bias"),                                                # bias q / k + contextual v
         # more than 51 buckets: the kernels' wide row pitches (csrc/irpe_attn.hip `Pitch`; everything above takes the narrow ones)
         ("qkv", True, 197, "euc", "ctx", 15.5), ("kv", False, 577, "quant", "ctx", 14.0), ("qk", False, 197, "euc", "bias", 13.0),
         ("qkv", True, 196, "euc", "ctx", 12.8),                                                # 51 buckets: the narrow pitches' limit
         ("qkv", True, 1025, "product"), ("kv", True, 2026, "euc", "ctx", 14.0)]                # long sequences (32 x 32 / 45 x 45 grids)


def _table(m):
    return m.lookup_table_bias if m.mode == "bias" else m.lookup_table_weight


@pytest.mark.parametrize("case", CASES, ids=lambda c: "-".join(str(x) for x in c))
def test_fused_irpe_attention_matches_restatement(case):
    from cream_amd import irpe as I, irpe_fused
    rpe_on, shared, L, method = case[:4]
    mode = case[4] if len(case) > 4 else "ctx"
    B, H = 2, 3
    torch.manual_seed(11)
    mods = [None, None, None]
    if rpe_on:
        kw = dict(ratio=case[5] if len(case) > 5 else 1.9, method=method, shared_head=shared, skip=0 if L == 196 else 1)
        cfg = I.get_rpe_config(mode=mode, rpe_on=rpe_on.replace("v", "") if mode == "bias" else rpe_on, **kw)
        if mode == "bias" and "v" in rpe_on:          # bias mode does not exist on the value side (irpe.py:468-470)
            cfg.rpe_v = I.get_rpe_config(mode="ctx", rpe_on="v", **kw).rpe_v
        mods = list(I.build_rpe(cfg, head_dim=64, num_heads=H))
    for m in mods:
        if m is not None:
            m.to(DEV)
            with torch.no_grad():
                _table(m).copy_(0.3 * torch.randn_like(_table(m)))
            _table(m).requires_grad_()
    qkv = (0.8 * torch.randn(B, L, 3, H, 64, device=DEV)).to(torch.bfloat16).requires_grad_()
    gy = torch.randn(B, L, H * 64, device=DEV).to(torch.bfloat16)
    assert irpe_fused.usable(qkv.dtype, qkv.device, 64, L, mods, False)
    y = irpe_fused.attention(qkv, 0.125, *mods)
    params = [_table(m) for m in mods if m is not None]
    got = torch.autograd.grad(y, [qkv] + params, gy)
    ref = _restatement(qkv, 0.125, mods)
    want = torch.autograd.grad(ref, [qkv] + params, gy.float())
    errs = dict(y=max_rel(y.float(), ref))
    for name, a, b in zip(["dq", "dk", "dv"], got[0].float().unbind(2), want[0].float().unbind(2)):
        errs[name] = max_rel(a, b)
    for name, a, b in zip([c for c, m in zip("qkv", mods) if m is not None], got[1:], want[1:]):
        errs["dW" + name] = max_rel(a.float(), b.float())
        assert a.shape == b.shape
    print(f"[fused irpe {rpe_on or 'none'} {method} {mode} shared={shared} L={L}]", {k: f"{v:.2e}" for k, v in errs.items()})
    assert all(torch.isfinite(t).all() for t in got)
    # product (50 buckets): 2x the worst measured (6.5e-3: bf16 P, dS and lookups).  euclidean / quant at ratio 1.9 have 8
    # buckets and ONE of them holds 85-95 % of all (query, key) pairs: its bucket gradient is a near-cancelling sum
    # (sum_j dS_ij = 0 for a softmax), which amplifies the same bf16 noise — measured up to 2.6e-2 on the rpe_k table
    bound = 1.3e-2 if method == "product" else 5e-2
    for k, v in errs.items():
        assert v < bound, (k, v, errs)


@pytest.mark.parametrize("case", [("k", True, 197, "ctx"), ("qkv", False, 197, "ctx"), ("qkv", True, 577, "ctx"), ("k", False, 196, "ctx"),
                                  ("qk", False, 197, "bias")], ids=lambda c: "-".join(str(x) for x in c))
def test_fused_cross_method_matches_rows_plus_cols(case):
    """iRPE_Cross (irpe.py:696-767: a rows iRPE + a cols iRPE, two gathers summed) on the fused kernels as ONE table over the
    occurring (row bucket, col bucket) pairs; the restatement evaluates the two modules separately, as the reference does, and
    the gradients are those of the four (six) parameters themselves."""
    from cream_amd import irpe as I, irpe_fused
    rpe_on, shared, L, mode = case
    B, H = 2, 3
    torch.manual_seed(5)
    cfg = I.get_rpe_config(ratio=1.9, method="cross", mode=mode, shared_head=shared, skip=0 if L == 196 else 1, rpe_on=rpe_on)
    mods = list(I.build_rpe(cfg, head_dim=64, num_heads=H))
    params = []
    for m in mods:
        if m is not None:
            assert type(m) is I.iRPE_Cross
            m.to(DEV)
            for part in (m.rp_rows, m.rp_cols):
                with torch.no_grad():
                    _table(part).copy_(0.3 * torch.randn_like(_table(part)))
                params.append(_table(part).requires_grad_())
    qkv = (0.8 * torch.randn(B, L, 3, H, 64, device=DEV)).to(torch.bfloat16).requires_grad_()
    gy = torch.randn(B, L, H * 64, device=DEV).to(torch.bfloat16)
    assert irpe_fused.usable(qkv.dtype, qkv.device, 64, L, mods, False)
    y = irpe_fused.attention(qkv, 0.125, *mods)
    got = torch.autograd.grad(y, [qkv] + params, gy)
    ref = _restatement(qkv, 0.125, mods)
    want = torch.autograd.grad(ref, [qkv] + params, gy.float())
    errs = dict(y=max_rel(y.float(), ref))
    for name, a, b in zip(["dq", "dk", "dv"], got[0].float().unbind(2), want[0].float().unbind(2)):
        errs[name] = max_rel(a, b)
    for i, (a, b) in enumerate(zip(got[1:], want[1:])):
        assert a.shape == b.shape
        errs[f"dW{i}"] = max_rel(a.float(), b.float())
    print(f"[fused irpe cross {rpe_on} {mode} shared={shared} L={L}]", {k: f"{v:.2e}" for k, v in errs.items()})
    assert all(torch.isfinite(t).all() for t in got)
    # rows / cols tables have 8 buckets each; a row (col) bucket gradient sums the near-cancelling dS of a whole stripe of keys
    # (the euclidean / quant bound of the test above)
    for k, v in errs.items():
        assert v < 5e-2, (k, v, errs)


def test_cross_module_takes_the_fused_path_under_autocast():
    from cream_amd import irpe as I, timing
    from cream_amd.rpe_attention import RPEAttention
    cfg = I.get_rpe_config(ratio=1.9, method="cross", mode="ctx", shared_head=True, skip=1, rpe_on="qkv")
    att = RPEAttention(192, num_heads=3, qkv_bias=True, rpe_config=cfg).to(DEV)
    x = torch.randn(2, 197, 192, device=DEV, requires_grad=True)
    timing.reset()
    timing.enable(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = att(x)
    y.float().sum().backward()
    timing.enable(False)
    names = set(timing.summary())
    assert {"irpe_attn_fwd", "irpe_attn_bwd"} <= names and not {"rpe_index_fwd", "rpe_index_bwd"} & names, names
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in att.parameters())


def test_module_takes_the_fused_path_under_autocast():
    """RPEAttention under bf16 autocast must run the fused kernels (timing regions prove which path ran)."""
    from cream_amd import irpe as I, timing
    from cream_amd.rpe_attention import RPEAttention
    cfg = I.get_rpe_config(ratio=1.9, method="product", mode="ctx", shared_head=True, skip=1, rpe_on="qkv")
    att = RPEAttention(192, num_heads=3, qkv_bias=True, rpe_config=cfg).to(DEV)
    x = torch.randn(2, 197, 192, device=DEV, requires_grad=True)
    timing.reset()
    timing.enable(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = att(x)
    y.float().sum().backward()
    timing.enable(False)
    names = set(timing.summary())
    assert {"irpe_attn_fwd", "irpe_attn_bwd"} <= names and not {"rpe_index_fwd", "rpe_index_bwd"} & names, names
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in att.parameters())


def _restatement_chunked(qkv, gy, scale, mods, chunk, round_lookups=True, ids=None):
    """The restatement above over batch chunks (an fp32 (B, H, L, L) map at config 4 is 1 GB and autograd keeps several):
    outputs and dq / dk / dv are per image, the table gradients are sums over images — accumulated in fp64."""
    params = [_table(m) for m in mods if m is not None]
    ys, dqkvs, dws = [], [], [torch.zeros_like(p, dtype=torch.float64) for p in params]
    for b0 in range(0, qkv.shape[0], chunk):
        x = qkv[b0:b0 + chunk].detach().clone().requires_grad_()
        y = _restatement(x, scale, mods, round_lookups, ids=ids)
        g = torch.autograd.grad(y, [x] + params, gy[b0:b0 + chunk].float())
        ys.append(y.detach())
        dqkvs.append(g[0].detach().float())
        for acc, t in zip(dws, g[1:]):
            acc += t.double()
    return torch.cat(ys), torch.cat(dqkvs), dws


@pytest.mark.parametrize("rpe_on", ["k", "qkv"])
def test_fused_irpe_attention_at_config4_matches_restatement(rpe_on):
    """BASELINE config 4 ITSELF — DeiT-B-384 + iRPE product / contextual, B = 64, H = 12, L = 577, 50 buckets, shared
    head (the shape bench.py's irpe_config4 leg times: 768 (b, h) items x 5 query tiles, per-(b, h) table-gradient
    products) — cream_irpe_attn_fwd / _bwd / _table_grad against the fp32 restatement evaluated in batch chunks:
    out, dq / dk / dv and the lookup-table gradients of every rpe present.  Bounds = 2x measured."""
    from cream_amd import irpe as I, irpe_fused
    B, H, L = 64, 12, 577
    torch.manual_seed(4)
    cfg = I.get_rpe_config(ratio=1.9, method="product", mode="ctx", shared_head=True, skip=1, rpe_on=rpe_on)
    mods = list(I.build_rpe(cfg, head_dim=64, num_heads=H))
    for m in mods:
        if m is not None:
            m.to(DEV)
            with torch.no_grad():
                _table(m).copy_(0.3 * torch.randn_like(_table(m)))
            _table(m).requires_grad_()
    qkv = (0.8 * torch.randn(B, L, 3, H, 64, device=DEV)).to(torch.bfloat16).requires_grad_()
    gy = torch.randn(B, L, H * 64, device=DEV).to(torch.bfloat16)
    assert irpe_fused.usable(qkv.dtype, qkv.device, 64, L, mods, False)
    y = irpe_fused.attention(qkv, 0.125, *mods)
    params = [_table(m) for m in mods if m is not None]
    got = torch.autograd.grad(y, [qkv] + params, gy)
    assert all(torch.isfinite(t).all() for t in got)
    names = ["dW" + c for c, m in zip("qkv", mods) if m is not None]
    # the restatement indexes with the ORACLE's bucket table (oracle/irpe_oracle.py, pinned against the reference-made
    # fixtures), not with ids taken from the product; the product's table must be that table
    from oracle import irpe_oracle
    oids, nb = irpe_oracle.product_bucket_ids(24, 24, skip=1, ratio=1.9)
    oids = torch.from_numpy(oids).to(DEV)
    for m in mods:
        if m is not None:
            assert m.num_buckets == nb and torch.equal(m.bucket_ids_for(L, DEV).long(), oids)
    report = {}
    for rounded in (True, False):
        ref_y, ref_dqkv, ref_dw = _restatement_chunked(qkv, gy, 0.125, mods, chunk=4, round_lookups=rounded, ids=oids)
        errs = dict(y=max_rel(y.float(), ref_y))
        for name, a, b in zip(["dq", "dk", "dv"], got[0].float().unbind(2), ref_dqkv.unbind(2)):
            errs[name] = max_rel(a, b)
        for name, a, b in zip(names, got[1:], ref_dw):
            assert a.shape == b.shape
            errs[name] = max_rel(a.double(), b)
        report[rounded] = errs
    print(f"[fused irpe config 4, rpe on {rpe_on}] vs restatement with bf16 lookups:", {k: f"{v:.2e}" for k, v in report[True].items()})
    print(f"[fused irpe config 4, rpe on {rpe_on}] vs pure fp32 restatement:       ", {k: f"{v:.2e}" for k, v in report[False].items()})
    # measured on the MI355X (profiles/r03_parity.txt): rpe on k: y 3.5e-3, dq 5.9e-3, dk 3.4e-3, dv 4.3e-3, dWk 3.6e-3;
    # rpe on q, k, v: y 5.8e-3, dq 8.2e-3, dk 4.9e-3, dv 4.2e-3, dWq 2.6e-3, dWk 3.6e-3, dWv 1.8e-3 — the same against the
    # pure fp32 restatement as against the one with bf16 lookups (the kernel's rounding of the lookups is not what bounds
    # the error), so BOTH are held to the same bounds = 2x the worst measured value of each quantity
    bound = dict(y=1.2e-2, dq=1.7e-2, dk=1.0e-2, dv=9e-3, dWq=7.5e-3, dWk=7.5e-3, dWv=7.5e-3)
    for rounded in (True, False):
        for k, v in report[rounded].items():
            assert v < bound[k], (k, v, rounded, report)


@pytest.mark.parametrize("L,B,H", [(77, 4, 8), (197, 2, 3), (33, 2, 1)])
def test_causal_attention_matches_masked_reference(L, B, H):
    """The `causal` switch of cream_irpe_attn_fwd / _bwd (keys j <= i only — the text towers' additive upper-triangular -inf
    mask, TinyCLIP/src/open_clip/model.py:756-762, without an (L, L) tensor) against masked softmax attention in fp32 on the
    same bf16 inputs: output, log-sum-exp, dq / dk / dv."""
    torch.manual_seed(L)
    qkv = (0.8 * torch.randn(B, L, 3, H, 64, device=DEV)).to(torch.bfloat16)
    gy = torch.randn(B, L, H * 64, device=DEV).to(torch.bfloat16)
    out, lse = irpe_fused.plain_fwd(qkv, 0.125, causal=True)
    dqkv = irpe_fused.plain_bwd(gy, qkv, out, lse, 0.125, causal=True)
    x = qkv.float().requires_grad_()
    q, k, v = x.permute(2, 0, 3, 1, 4).unbind(0)
    a = (q * 0.125) @ k.transpose(-2, -1) + torch.full((L, L), float("-inf"), device=DEV).triu_(1)
    ref = (a.softmax(-1) @ v).transpose(1, 2).reshape(B, L, H * 64)
    (rg,) = torch.autograd.grad(ref, x, gy.float())
    errs = dict(y=max_rel(out.float(), ref), lse=max_rel(lse, torch.logsumexp(a, -1)),
                **{n: max_rel(g_, r_) for n, g_, r_ in zip(("dq", "dk", "dv"), dqkv.float().unbind(2), rg.unbind(2))})
    print(f"[causal attention L={L}]", {k_: f"{v_:.2e}" for k_, v_ in errs.items()})
    assert all(v_ < 1.3e-2 for v_ in errs.values()), errs


from cream_amd import irpe_fused  # noqa: E402


@pytest.mark.parametrize("rpe_on,rate,L", [("", 0.1, 197), ("qkv", 0.1, 197), ("kv", 0.5, 50), ("k", 0.25, 577)])
def test_fused_attention_dropout_matches_masked_restatement(rpe_on, rate, L):
    """Attention dropout inside the fused kernels (rpe_vision_transformer.py:86: the softmax output is dropped BEFORE the
    value product and the value-side bucket sums): the keep mask is a counter-based hash of (seed, b, h, i, j), restated in
    numpy by irpe_fused.dropout_keep_mask — forward and every gradient against the fp32 restatement with THAT mask; the
    kept fraction is the requested one; another seed gives another mask, the same seed the same bits."""
    import numpy as np
    from cream_amd import irpe as I, irpe_fused
    B, H = 2, 3
    torch.manual_seed(21)
    mods = [None, None, None]
    if rpe_on:
        cfg = I.get_rpe_config(ratio=1.9, method="product", mode="ctx", shared_head=True, skip=1, rpe_on=rpe_on)
        mods = list(I.build_rpe(cfg, head_dim=64, num_heads=H))
    for m in mods:
        if m is not None:
            m.to(DEV)
            with torch.no_grad():
                _table(m).copy_(0.3 * torch.randn_like(_table(m)))
            _table(m).requires_grad_()
    qkv = (0.8 * torch.randn(B, L, 3, H, 64, device=DEV)).to(torch.bfloat16).requires_grad_()
    gy = torch.randn(B, L, H * 64, device=DEV).to(torch.bfloat16)
    seed = 12345
    hashes = irpe_fused.dropout_keep_mask(seed, B, H, L)
    keep_np = hashes >= np.uint32(irpe_fused.dropout_threshold(rate))
    assert abs(keep_np.mean() - (1 - rate)) < 4.5 * (rate * (1 - rate) / keep_np.size) ** 0.5, keep_np.mean()     # 4.5 sigma
    assert (irpe_fused.dropout_keep_mask(seed + 1, B, H, L) >= np.uint32(irpe_fused.dropout_threshold(rate))).mean() != keep_np.mean()
    keep = torch.from_numpy(keep_np).to(DEV).float() / (1.0 - float(np.float32(rate)))
    y = irpe_fused.attention(qkv, 0.125, *mods, dropout_p=rate, seed=seed)
    y2 = irpe_fused.attention(qkv, 0.125, *mods, dropout_p=rate, seed=seed)
    y3 = irpe_fused.attention(qkv, 0.125, *mods, dropout_p=rate, seed=seed + 1)
    assert torch.equal(y, y2) and not torch.equal(y, y3)
    params = [_table(m) for m in mods if m is not None]
    got = torch.autograd.grad(y, [qkv] + params, gy)
    ref = _restatement(qkv, 0.125, mods, keep=keep)
    want = torch.autograd.grad(ref, [qkv] + params, gy.float())
    errs = dict(y=max_rel(y.float(), ref))
    for name, a, b in zip(["dq", "dk", "dv"], got[0].float().unbind(2), want[0].float().unbind(2)):
        errs[name] = max_rel(a, b)
    for name, a, b in zip([c for c, m in zip("qkv", mods) if m is not None], got[1:], want[1:]):
        errs["dW" + name] = max_rel(a.float(), b.float())
    print(f"[fused irpe dropout {rpe_on or 'none'} p={rate} L={L}]", {k: f"{v:.2e}" for k, v in errs.items()})
    assert all(torch.isfinite(t).all() for t in got)
    for k, v in errs.items():
        assert v < 1.5e-2, (k, v, errs)
    # without the mask the restatement is far away: the comparison above is a statement about the mask
    assert max_rel(y.float(), _restatement(qkv, 0.125, mods).detach()) > 5e-2


def test_module_with_attention_dropout_stays_on_the_fused_kernels():
    from cream_amd import irpe as I, timing
    from cream_amd.rpe_attention import RPEAttention
    cfg = I.get_rpe_config(ratio=1.9, method="product", mode="ctx", shared_head=True, skip=1, rpe_on="k")
    att = RPEAttention(192, num_heads=3, qkv_bias=True, attn_drop=0.1, rpe_config=cfg).to(DEV)
    x = torch.randn(2, 197, 192, device=DEV, requires_grad=True)
    timing.reset()
    timing.enable(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = att(x)
    y.float().sum().backward()
    timing.enable(False)
    names = set(timing.summary())
    assert {"irpe_attn_fwd", "irpe_attn_bwd"} <= names and not {"rpe_index_fwd", "rpe_index_bwd"} & names, names
    att.eval()
    with torch.autocast("cuda", dtype=torch.bfloat16), torch.no_grad():
        a, b = att(x), att(x)
    assert torch.equal(a, b)                                   # no dropout outside training

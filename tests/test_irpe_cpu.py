"""iRPE host logic on CPU against golden vectors produced by the reference itself
(tests/golden/irpe_*.npz|json, deit_tiny_irpe_k.npz; generator: tests/golden/make_golden.py):
bucket ids and the piecewise index are integer work -> bit-exact; module outputs/gradients and
the DeiT-tiny forward (BASELINE config 1) within fp32 round-off."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT
from helpers import load_json, load_npz, max_rel

sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from fixture_utils import fill_params  # noqa: E402

from cream_amd import irpe as I  # noqa: E402
import refshim  # noqa: E402

METH = dict(product=I.METHOD.PRODUCT, euc=I.METHOD.EUCLIDEAN, quant=I.METHOD.QUANT,
            cross_rows=I.METHOD.CROSS_ROWS, cross_cols=I.METHOD.CROSS_COLS)


def test_bucket_tables_bit_exact():
    fix, meta = load_npz("irpe_buckets.npz"), load_json("irpe_buckets.json")
    assert len(meta) == 12
    for m in meta:
        r = m["ratio"]
        ids, nb = I.get_bucket_ids_2d(METH[m["method"]], m["h"], m["w"], m["skip"], r, 2 * r, 8 * r, dtype=torch.long)
        assert nb == m["num_buckets"] and list(ids.shape) == m["shape"] and int(ids.sum()) == m["sum"], m["key"]
        if m["key"] in fix:
            np.testing.assert_array_equal(ids.numpy(), fix[m["key"]].astype(np.int64))
        else:
            np.testing.assert_array_equal(ids.numpy()[::16], fix[m["key"] + "|rows16"].astype(np.int64))
    # SURVEY §4 known answers: product, ratio 1.9, skip 1
    ids, nb = I.get_bucket_ids_2d(I.METHOD.PRODUCT, 14, 14, 1, 1.9, 3.8, 15.2)
    assert nb == 50 and int(ids.sum()) == 941241 and ids[1, 1:8].tolist() == [24, 23, 22, 22, 21, 21, 21]
    assert int(I.get_bucket_ids_2d(I.METHOD.PRODUCT, 24, 24, 1, 1.9, 3.8, 15.2)[0].sum()) == 8019121


def test_piecewise_index_bit_exact():
    fix = load_npz("irpe_buckets.npz")
    xs = torch.arange(-64, 65)
    xf = torch.arange(0, 400).float().sqrt().round()
    for ratio in (1.9, 3.0, 7.5, 20, 51):
        key = f"{ratio}"
        np.testing.assert_array_equal(I.piecewise_index(xs, ratio, 2 * ratio, 8 * ratio, torch.long).numpy(),
                                      fix[f"piecewise_int_{key}"].astype(np.int64))
        np.testing.assert_array_equal(I.piecewise_index(xf, ratio, 2 * ratio, 8 * ratio, torch.long).numpy(),
                                      fix[f"piecewise_flt_{key}"].astype(np.int64))


def test_config_builders():
    c = I.get_rpe_config(ratio=1.9, method="product", mode="ctx", shared_head=True, skip=1, rpe_on="qkv")
    assert c.rpe_k.alpha == 1.9 and c.rpe_k.beta == 3.8 and c.rpe_k.gamma == 15.2 and c.rpe_k.num_buckets == 50
    q, k, v = I.build_rpe(c, head_dim=64, num_heads=3)
    assert q.transposed and k.transposed and not v.transposed
    assert tuple(k.lookup_table_weight.shape) == (1, 64, 50) and tuple(v.lookup_table_weight.shape) == (1, 50, 64)
    c2 = I.get_rpe_config(ratio=1.9, method="cross", mode="bias", shared_head=False, skip=0, rpe_on="k")
    _, k2, _ = I.build_rpe(c2, head_dim=32, num_heads=4)
    assert isinstance(k2, I.iRPE_Cross) and tuple(k2.rp_rows.lookup_table_bias.shape) == (4, 7)
    assert I.build_rpe(None, 64, 3) == (None, None, None)
    with pytest.raises(NotImplementedError):
        I.iRPE(64, 3, mode="bias", method=I.METHOD.PRODUCT, transposed=False, num_buckets=50)


@pytest.mark.parametrize("tag,kw", [("ctx_shared", dict(mode="ctx", shared_head=True)),
                                    ("ctx_perhead", dict(mode="ctx", shared_head=False)),
                                    ("bias_perhead", dict(mode="bias", shared_head=False))])
def test_modules_match_reference_outputs_and_grads(tag, kw):
    fix = load_npz("irpe_modules.npz")
    cfg = I.get_rpe_config(ratio=1.9, method="product", skip=1, rpe_on="qkv" if kw["mode"] == "ctx" else "qk", **kw)
    mods = I.build_rpe(cfg, head_dim=64, num_heads=3)
    g = torch.Generator().manual_seed(31)
    for which, mod in zip("qkv", mods):
        if mod is None:
            continue
        with torch.no_grad():
            for p in mod.parameters():
                p.copy_(0.3 * torch.randn(p.shape, generator=g))
        if which == "v":
            x = torch.randn(2, 3, 197, 197, generator=g).softmax(-1).requires_grad_()
        else:
            x = torch.randn(2, 3, 197, 64, generator=g, requires_grad=True)
        y = mod(x)
        gy = torch.randn(y.shape, generator=g)
        grads = torch.autograd.grad(y, [x] + list(mod.parameters()), gy, allow_unused=True)
        ysub = y[:, :, ::7, ::5] if y.shape[-1] == 197 else y[:, :, ::7]
        assert max_rel(ysub.detach(), fix[f"{tag}|{which}|y"]) < 1e-5
        assert abs(float(y.double().sum()) - float(fix[f"{tag}|{which}|ysum"][0])) < 1e-3 * max(1.0, abs(float(fix[f"{tag}|{which}|ysum"][0])))
        if grads[0] is not None:
            dsub = grads[0][:, :, ::7, ::5] if grads[0].shape[-1] == 197 else grads[0][:, :, ::7]
            assert max_rel(dsub, fix[f"{tag}|{which}|dx"]) < 1e-5
        assert max_rel(grads[1], fix[f"{tag}|{which}|dw"]) < 1e-5


def test_rpe_attention_matches_reference():
    from cream_amd.rpe_attention import RPEAttention
    fix = load_npz("irpe_attention.npz")
    cfg = I.get_rpe_config(ratio=1.9, method="product", mode="ctx", shared_head=True, skip=1, rpe_on="qkv")
    att = RPEAttention(192, num_heads=3, qkv_bias=True, rpe_config=cfg)
    fill_params(att, seed=19)
    with torch.no_grad():
        for n, p in att.named_parameters():
            if "lookup_table" in n:
                p.copy_(0.3 * torch.randn(p.shape, generator=torch.Generator().manual_seed(len(n))))
    g = torch.Generator().manual_seed(41)
    x = torch.randn(2, 197, 192, generator=g, requires_grad=True)
    gy = torch.randn(2, 197, 192, generator=g)
    y = att(x)
    y.backward(gy)
    assert max_rel(y.detach(), fix["y"]) < 1e-5 and max_rel(x.grad, fix["dx"]) < 1e-5
    for k, v in fix.items():
        if k.startswith("full|"):
            assert max_rel(dict(att.named_parameters())[k[5:]].grad, v) < 1e-5, k


def test_deit_tiny_irpe_k_single_image_forward():
    """BASELINE config 1: DeiT-tiny + iRPE (contextual product, 50 buckets, shared head, on keys)."""
    from cream_amd.rpe_attention import deit_tiny_patch16_224_ctx_product_50_shared_k
    fix = load_npz("deit_tiny_irpe_k.npz")
    torch.manual_seed(0)
    model = deit_tiny_patch16_224_ctx_product_50_shared_k()
    assert sum(p.numel() for p in model.parameters()) == int(fix["n_params"][0]) == 5755816
    fill_params(model, seed=17)
    model.eval()
    x = torch.randn(1, 3, 224, 224, generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        logits = model(x)
    assert max_rel(logits, fix["logits"]) < 1e-4


def _ext_cases():
    import json
    return json.load(open(os.path.join(ROOT, "tests", "golden", "irpe_modules_ext.json")))


def _zseed(tag):
    import zlib
    return zlib.crc32(tag.encode()) & 0x7fffffff


def run_ext_case(I, case, device="cpu", tol=1e-5):
    """One case of the reference-made irpe_modules_ext fixture (bias / euclidean / quant / cross,
    skip = 0, non-square maps) through the module family `I` on `device`."""
    fix = load_npz("irpe_modules_ext.npz")
    tag, kw, heads, h, w, L = case["tag"], case["kw"], case["heads"], case["h"], case["w"], case["L"]
    cfg = I.get_rpe_config(**kw)
    mods = I.build_rpe(cfg, head_dim=64, num_heads=heads)
    g = torch.Generator().manual_seed(_zseed(tag))
    for which, mod in zip("qkv", mods):
        if mod is None:
            continue
        with torch.no_grad():
            for p in mod.parameters():
                p.copy_(0.3 * torch.randn(p.shape, generator=g))
        if which == "v":
            x = torch.randn(2, heads, L, L, generator=g).softmax(-1)
        else:
            x = torch.randn(2, heads, L, 64, generator=g)
        mod = mod.to(device)
        x = x.to(device).requires_grad_()
        y = mod(x, height=h, width=w)
        gy = torch.randn(y.shape, generator=g).to(device)
        params = list(mod.parameters())
        grads = torch.autograd.grad(y, [x] + params, gy, allow_unused=True)
        sub = (lambda t: t[:, :, ::7, ::5] if (t.shape[-1] == L and L > 150) else t)
        assert max_rel(sub(y).detach().cpu(), fix[f"{tag}|{which}|y"]) < tol, (tag, which)
        ysum = float(fix[f"{tag}|{which}|ysum"][0])
        assert abs(float(y.double().sum()) - ysum) < 1e-3 * max(1.0, abs(ysum))
        if grads[0] is not None:
            assert max_rel(sub(grads[0]).cpu(), fix[f"{tag}|{which}|dx"]) < tol, (tag, which)
        for i, gp in enumerate(grads[1:]):
            assert max_rel(gp.cpu(), fix[f"{tag}|{which}|dw{i}"]) < tol, (tag, which, i)
        mod.to("cpu")


@pytest.mark.parametrize("case", _ext_cases(), ids=lambda c: c["tag"])
def test_extended_family_matches_reference(case):
    run_ext_case(I, case)


@pytest.mark.parametrize("rpe_on", ["k", "qkv"])
def test_rpe_attention_L577_matches_reference(rpe_on):
    run_attention_L577(rpe_on, "cpu", 1e-4)


def run_attention_L577(rpe_on, device, tol, autocast=False, guard=None):
    from cream_amd.rpe_attention import RPEAttention
    fix = load_npz("irpe_attention_L577.npz")
    cfg = I.get_rpe_config(ratio=1.9, method="product", mode="ctx", shared_head=True, skip=1, rpe_on=rpe_on)
    att = RPEAttention(192, num_heads=3, qkv_bias=True, rpe_config=cfg)
    fill_params(att, seed=29)
    with torch.no_grad():
        for n, p in att.named_parameters():
            if "lookup_table" in n:
                p.copy_(0.3 * torch.randn(p.shape, generator=torch.Generator().manual_seed(len(n) + 5)))
    att = att.to(device)
    g = torch.Generator().manual_seed(43)
    x = torch.randn(1, 577, 192, generator=g).to(device).requires_grad_()
    gy = torch.randn(1, 577, 192, generator=g).to(device)
    import contextlib
    with (guard if guard is not None else contextlib.nullcontext()):
        with torch.autocast(torch.device(device).type, dtype=torch.bfloat16, enabled=autocast):
            y = att(x)
        y.float().backward(gy)
    worst = max(max_rel(y[:, ::3].detach().float().cpu(), fix[f"{rpe_on}|y"]), max_rel(x.grad[:, ::3].cpu(), fix[f"{rpe_on}|dx"]))
    for k, v in fix.items():
        if k.startswith(f"{rpe_on}|full|"):
            worst = max(worst, max_rel(dict(att.named_parameters())[k[len(rpe_on) + 6:]].grad.cpu(), v))
    assert worst < tol, worst
    return worst


@pytest.mark.skipif(not refshim.have_reference(), reason="needs the reference checkout")
def test_reference_rpe_attention_through_the_patched_caller():
    """The reference's UNCHANGED rpe_vision_transformer.py imported next to the drop-in `irpe`, its RPEAttention
    patched onto our forward (dropin.install_irpe): same parameters, same inputs -> the reference-made fixture.
    (On the CPU the patched forward takes the composed path; on the device under bf16 autocast the same patched
    class runs the fused kernels — tests/test_irpe_gpu.py exercises that forward through cream_amd.rpe_attention.)"""
    import cream_amd.dropin as d
    refshim._install_easydict()
    refshim._install_timm_stub()
    caller = d.install_irpe(refshim.IRPE)
    try:
        assert caller.RPEAttention._cream_fast_path and caller.__file__.startswith(refshim.IRPE)
        irpe = sys.modules["irpe"]
        assert os.path.dirname(irpe.__file__) == d.PATH                      # the drop-in, not the reference's
        fix = load_npz("irpe_attention.npz")
        cfg = irpe.get_rpe_config(ratio=1.9, method="product", mode="ctx", shared_head=True, skip=1, rpe_on="qkv")
        att = caller.RPEAttention(192, num_heads=3, qkv_bias=True, rpe_config=cfg)
        fill_params(att, seed=19)
        with torch.no_grad():
            for n, p in att.named_parameters():
                if "lookup_table" in n:
                    p.copy_(0.3 * torch.randn(p.shape, generator=torch.Generator().manual_seed(len(n))))
        g = torch.Generator().manual_seed(41)
        x = torch.randn(2, 197, 192, generator=g, requires_grad=True)
        gy = torch.randn(2, 197, 192, generator=g)
        y = att(x)
        y.backward(gy)
        assert max_rel(y.detach(), fix["y"]) < 1e-5 and max_rel(x.grad, fix["dx"]) < 1e-5
        # the model's block loop is patched as well (on the device under bf16 autocast: one node for all RPEBlocks,
        # cream_amd/deit_native.py); where that node does not apply — here — it is the caller's own loop
        assert caller.VisionTransformer._cream_fast_path
        cfg_k = irpe.get_rpe_config(ratio=1.9, method="product", mode="ctx", shared_head=True, skip=1, rpe_on="k")
        model = caller.VisionTransformer(img_size=64, patch_size=16, embed_dim=192, depth=2, num_heads=3, num_classes=10, qkv_bias=True,
                                         rpe_config=cfg_k)
        img = torch.randn(2, 3, 64, 64, generator=g)
        with torch.no_grad():
            assert torch.equal(model.forward_features(img), model._cream_reference_forward_features(img))
    finally:
        if refshim.IRPE in sys.path:
            sys.path.remove(refshim.IRPE)
        for k in [k for k in sys.modules if k in ("irpe", "rpe_vision_transformer", "rpe_index_cpp") or k.startswith("rpe_ops")]:
            del sys.modules[k]


def test_irpe_oracle_is_pinned():
    """oracle/irpe_oracle.py (the reference's pure-PyTorch RPEAttention formulation, used as the CPU baseline of bench.py's
    config-4 leg) against reference-made fixtures: the product bucket tables (full 14x14, sampled rows + checksum of
    24x24) and an RPEAttention forward / backward with rpe on q, k and v."""
    from oracle import irpe_oracle as IO
    z = load_npz("irpe_buckets.npz")
    meta = {m["key"]: m for m in load_json("irpe_buckets.json")}
    ids, nb = IO.product_bucket_ids(14, 14, 1)
    assert nb == meta["product_1.9_14x14_s1"]["num_buckets"] and np.array_equal(ids, z["product_1.9_14x14_s1"])
    ids577, nb577 = IO.product_bucket_ids(24, 24, 1)
    assert nb577 == 50 and int(ids577.sum()) == meta["product_1.9_24x24_s1"]["sum"] == 8019121
    assert np.array_equal(ids577[::16], z["product_1.9_24x24_s1|rows16"])
    ids70, nb70 = IO.product_bucket_ids(7, 10, 0)
    assert nb70 == 49 and np.array_equal(ids70, z["product_1.9_7x10_s0"])
    ids145, _ = IO.product_bucket_ids(12, 12, 1, ratio=3.0)
    assert np.array_equal(ids145, z["product_3.0_12x12_s1"])

    fix = load_npz("irpe_attention.npz")
    att = torch.nn.ModuleDict(dict(qkv=torch.nn.Linear(192, 576), proj=torch.nn.Linear(192, 192)))
    params = {"qkv.weight": None, "qkv.bias": None, "rpe_q.lookup_table_weight": (1, 64, 50), "rpe_k.lookup_table_weight": (1, 64, 50),
              "rpe_v.lookup_table_weight": (1, 50, 64), "proj.weight": None, "proj.bias": None}
    # the parameter names (and their order) of the reference's RPEAttention: fill_params keys its seeds on them
    holder = {}
    for name, shape in params.items():
        holder[name] = dict(att.named_parameters())[name].detach().clone() if shape is None else torch.zeros(shape)
    fill_params(holder, seed=19)
    for n in holder:
        if "lookup_table" in n:
            holder[n] = 0.3 * torch.randn(holder[n].shape, generator=torch.Generator().manual_seed(len(n)))
    holder = {k: v.requires_grad_() for k, v in holder.items()}
    g = torch.Generator().manual_seed(41)
    x = torch.randn(2, 197, 192, generator=g, requires_grad=True)
    gy = torch.randn(2, 197, 192, generator=g)
    y = IO.rpe_attention_layer(holder, x, 3, ids, nb)
    y.backward(gy)
    assert max_rel(y.detach(), fix["y"]) < 1e-5 and max_rel(x.grad, fix["dx"]) < 1e-5
    for k, v in fix.items():
        if k.startswith("full|"):
            assert max_rel(holder[k[5:]].grad, v) < 1e-5, k


def test_model_zoo_checkpoints_fit():
    """The six checkpoints of the reference's model zoo (rpe_models.py:10-19): the constructors of the same names produce
    exactly the state-dict keys, shapes and parameter counts of the reference's classes (fixture irpe_zoo.json, generated by
    instantiating them), a published-format file ({'model': state_dict}) loads strictly through the tensors-only reader,
    and an unknown name fails like the reference's assertion."""
    import cream_amd.rpe_attention as R
    zoo = load_json("irpe_zoo.json")
    assert set(zoo) == set(R.PROVIDED_CHECKPOINTS)
    for name, rec in zoo.items():
        m = getattr(R, name)()
        sd = m.state_dict()
        assert list(sd.keys()) == rec["keys"], name
        assert [list(v.shape) for v in sd.values()] == rec["shapes"], name
        assert sum(p.numel() for p in m.parameters()) == rec["n_params"], name
    import tempfile
    name = "deit_small_patch16_224_ctx_product_50_shared_qkv"
    src = getattr(R, name)()
    with torch.no_grad():
        for p in src.parameters():
            p.add_(0.01)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, name + ".pth")
        torch.save({"model": src.state_dict()}, path)
        dst = R.create_zoo_model(name, path)
    assert all(torch.equal(a, b) for a, b in zip(src.state_dict().values(), dst.state_dict().values()))
    with pytest.raises(AssertionError, match="not provided"):
        R.create_zoo_model("deit_tiny_patch16_224_ctx_product_50_shared_qkv")
    # DeiT-style files: an argparse.Namespace next to the weights loads; arbitrary objects are refused with advice
    import argparse
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, name + ".pth")
        torch.save({"model": src.state_dict(), "args": argparse.Namespace(lr=1e-3, model=name), "epoch": 3}, path)
        dst = R.create_zoo_model(name, path)
        assert all(torch.equal(a, b) for a, b in zip(src.state_dict().values(), dst.state_dict().values()))
        torch.save({"model": src.state_dict(), "sched": torch.optim.lr_scheduler.LambdaLR}, path)
        with pytest.raises(RuntimeError, match="pass the dictionary"):
            R.create_zoo_model(name, path)


def test_dropout_keep_mask_restatement_is_deterministic_and_unbiased():
    """The numpy restatement of the kernels' keep mask (csrc/irpe_attn.hip drop_key / drop_keep): a pure function of
    (seed, b, h, i, j), the requested keep fraction, independent between heads and seeds."""
    import numpy as np
    from cream_amd import irpe_fused
    h1 = irpe_fused.dropout_keep_mask(7, 2, 3, 64)
    assert h1.shape == (2, 3, 64, 64) and h1.dtype == np.uint32
    assert np.array_equal(h1, irpe_fused.dropout_keep_mask(7, 2, 3, 64))
    assert np.array_equal(h1[:, :, :32, :32], irpe_fused.dropout_keep_mask(7, 2, 3, 32))      # does not depend on L
    for rate in (0.1, 0.5, 0.9):
        keep = h1 >= np.uint32(irpe_fused.dropout_threshold(rate))
        assert abs(keep.mean() - (1 - rate)) < 4.5 * (rate * (1 - rate) / keep.size) ** 0.5
    other = irpe_fused.dropout_keep_mask(8, 2, 3, 64)
    agree = ((h1 >> 31) == (other >> 31)).mean()                                               # top bits: 50 % agreement
    assert 0.47 < agree < 0.53
    assert 0.47 < ((h1[0, 0] >> 31) == (h1[0, 1] >> 31)).mean() < 0.53
    assert irpe_fused.dropout_threshold(0.5) == 2 ** 31 and irpe_fused.dropout_threshold(1e-12) == 1


@pytest.mark.parametrize("mode,rpe_on,L", [("ctx", "kv", 197), ("bias", "k", 196), ("ctx", "q", 50)])
def test_cross_one_table_view_equals_rows_plus_cols(mode, rpe_on, L):
    """iRPE_Cross.merged_table / merged_ids_for (the operand the fused kernels take for the cross method): one lookup over
    the occurring (row bucket, col bucket) pairs is rows + cols of irpe.py:758-760, values and parameter gradients."""
    torch.manual_seed(3)
    cfg = I.get_rpe_config(ratio=1.9, method="cross", mode=mode, shared_head=False, skip=0 if L == 196 else 1, rpe_on=rpe_on)
    for m in I.build_rpe(cfg, head_dim=16, num_heads=2):
        if m is None:
            continue
        params = [p for p in m.parameters()]
        with torch.no_grad():
            for p in params:
                p.normal_()
        ids, ir, ic, nb = m.merged_ids_for(L, "cpu")
        assert ids.dtype == torch.int32 and ids.shape == (L, L) and int(ids.max()) == nb - 1 and nb <= 64
        table = m.merged_table(L, "cpu")
        x = torch.randn(2, 2, L, 16 if m.transposed else L)
        if not m.transposed:
            x = x.softmax(-1)
        want = m(x)
        if mode == "bias":
            got = table[:, ids.flatten().long()].view(1, 2, L, L)
        elif m.transposed:
            got = (x @ table.unsqueeze(0)).gather(-1, ids.long().expand(2, 2, L, L))
        else:
            sums = torch.zeros(2, 2, L, nb).scatter_add_(-1, ids.long().expand(2, 2, L, L), x)
            got = sums @ table.unsqueeze(0)
        assert torch.allclose(got, want, rtol=1e-4, atol=1e-4)
        g = torch.randn_like(want)
        for a, b in zip(torch.autograd.grad(got, params, g), torch.autograd.grad(want, params, g)):
            assert torch.allclose(a, b, rtol=1e-3, atol=1e-3)

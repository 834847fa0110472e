"""AutoFormer hot path on the CPU: (1) the oracle is pinned against golden vectors that the
reference itself produced; (2) the product's host-side modules (same classes that run on
the GPU, here on host tensors through the C ABI's *_host entry points) reproduce them."""
import os
import random
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT
from helpers import check_against_fixture, config_of, load_json, load_npz, max_rel

sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from fixture_utils import SUBNET_S, SUBNET_T, SUPERNETS, fill_params, make_batch, model_kwargs  # noqa: E402

from oracle import autoformer_oracle as AO  # noqa: E402

KAT = load_json("autoformer_kat.json")


# ---------------------------------------------------------------- oracle pinned to the reference
def test_oracle_sample_configs_golden_draws():
    for size in ("T", "S"):
        for epoch in (0, 1, 7):
            random.seed(epoch)
            got = [AO.sample_configs(SUPERNETS[size]["choices"]) for _ in range(3)]
            assert got == KAT[f"draws_{size}_epoch{epoch}"]
    # SURVEY Appendix C.1
    random.seed(0)
    c = AO.sample_configs(SUPERNETS["S"]["choices"])
    assert c["layer_num"] == 13 and c["embed_dim"][0] == 448
    assert c["num_heads"] == [6, 5, 5, 7, 6, 7, 7, 7, 5, 6, 5, 7, 5]


def test_oracle_rel_index_tables():
    fix = load_npz("autoformer_rel_index.npz")
    _, fv, fh = AO.rel_pos_embeddings(torch.zeros(30, 4), torch.zeros(30, 4), 197, 14)
    assert np.array_equal(fv.numpy(), fix["iv"]) and np.array_equal(fh.numpy(), fix["ih"])
    assert int(fv.sum()) == KAT["rel_index_sum_v"] == 576240
    assert int(fh.sum()) == KAT["rel_index_sum_h"] == 576240


@pytest.mark.parametrize("size,batch", [("T", 2), ("S", 1)])
def test_oracle_step_matches_reference(size, batch):
    fix = load_npz(f"autoformer_{size}_step.npz")
    cfg = config_of(fix)
    from cream_amd.autoformer import Vision_TransformerSuper
    m = Vision_TransformerSuper(**model_kwargs(size))       # only used as a named-parameter container
    fill_params(m, seed=3)
    sd = {k: v.detach() for k, v in m.named_parameters()}
    images, target = make_batch(batch, seed=5)
    loss, grads = AO.train_step(sd, cfg, images, target)
    with torch.no_grad():
        logits = AO.forward(sd, cfg, images)
    worst = check_against_fixture(fix, logits, loss, grads, tol=2e-5)
    assert worst < 2e-5


def test_oracle_attention_matches_reference():
    fix = load_npz("autoformer_attention.npz")
    from cream_amd.autoformer import AttentionSuper
    att = AttentionSuper(256, num_heads=4, qkv_bias=True, relative_position=True, change_qkv=True)
    fill_params(att, seed=11)
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in att.named_parameters()}
    g = torch.Generator().manual_seed(21)
    x = torch.randn(2, 197, 216, generator=g, requires_grad=True)
    gy = torch.randn(2, 197, 216, generator=g)
    y = AO.attention(sd, "", x, 216, 3)
    y.backward(gy)
    assert max_rel(y.detach(), fix["y"]) < 1e-5
    assert max_rel(x.grad, fix["dx"]) < 1e-5
    for k, v in fix.items():
        if k.startswith("full|"):
            assert max_rel(sd[k[5:]].grad, v) < 1e-5, k


# ---------------------------------------------------------------- product host logic
def test_param_count_kats():
    """README sizes 5.8M / 22.9M / 53.7M (AutoFormer/README.md:60-62) and supernet totals."""
    from cream_amd.autoformer import Vision_TransformerSuper
    for size, sub, want in (("T", SUBNET_T, 5867944), ("S", SUBNET_S, 22891432)):
        m = Vision_TransformerSuper(**model_kwargs(size))
        assert m.get_sampled_params_numel(sub) == want == KAT[f"subnet_{size}_params"]
        assert sum(p.numel() for p in m.parameters()) == KAT[f"supernet_{size}_params"]
        assert sorted(m.state_dict().keys()) == KAT[f"supernet_{size}_state_keys"]     # checkpoint compatible
        assert abs(m.get_complexity(196) - KAT[f"supernet_{size}_complexity"]) < 1e-3 * KAT[f"supernet_{size}_complexity"]
    mB = Vision_TransformerSuper(**model_kwargs("B"))
    assert sum(p.numel() for p in mB.parameters()) == KAT["supernet_B_params"] == 80160360


def test_engine_sample_configs_golden_draws():
    from cream_amd.autoformer import engine
    for size in ("T", "S"):
        for epoch in (0, 1, 7):
            random.seed(epoch)
            got = [engine.sample_configs(engine.SEARCH_SPACES[size]["choices"]) for _ in range(3)]
            assert got == KAT[f"draws_{size}_epoch{epoch}"]


def test_relative_index_tables_bit_exact():
    from cream_amd.autoformer.modules import relative_index_tables
    fix = load_npz("autoformer_rel_index.npz")
    iv, ih = relative_index_tables(197, 14)
    assert iv.dtype == torch.int32 and iv.is_contiguous()
    assert np.array_equal(iv.numpy(), fix["iv"]) and np.array_equal(ih.numpy(), fix["ih"])
    # rows 1 and 29 of each table are never used with a 14x14 grid
    used = set(iv.unique().tolist())
    assert used == {0} | set(range(2, 29))


def test_qkv_interleave_and_contiguous_bias():
    """qkv_super.py:72-83: rows 0,3,6.. -> q etc., bias is the plain prefix."""
    from cream_amd.autoformer import qkv_super
    lin = qkv_super(8, 12, bias=True)
    with torch.no_grad():
        lin.weight.copy_(torch.arange(96.).view(12, 8))
        lin.bias.copy_(torch.arange(12.))
    lin.set_sample_config(sample_in_dim=5, sample_out_dim=6)
    w, b = lin.samples["weight"], lin.samples["bias"]
    assert torch.equal(w, torch.stack([lin.weight[r, :5] for r in (0, 3, 1, 4, 2, 5)]))
    assert torch.equal(b, lin.bias[:6])
    assert lin.calc_sampled_param_num() == 36


@pytest.mark.parametrize("size,batch", [("T", 2), ("S", 1)])
def test_product_step_on_host_matches_reference(size, batch):
    """Whole supernet step through the product's modules on HOST tensors (bucketed
    attention, C-ABI host entry points), fp32, against the reference's golden step."""
    fix = load_npz(f"autoformer_{size}_step.npz")
    cfg = config_of(fix)
    from cream_amd.autoformer import Vision_TransformerSuper
    m = Vision_TransformerSuper(**model_kwargs(size))
    fill_params(m, seed=3)
    m.set_sample_config(cfg)
    m.train()
    images, target = make_batch(batch, seed=5)
    logits = m(images)
    loss = AO.soft_target_cross_entropy(logits, target)
    loss.backward()
    grads = {k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in m.named_parameters()}
    worst = check_against_fixture(fix, logits, loss, grads, tol=1e-3)
    assert worst < 1e-3
    # structural: gradients are exactly zero outside the sampled slice
    E = cfg["embed_dim"][0]
    g = m.blocks[0].attn.qkv.weight.grad
    assert torch.count_nonzero(g[:, E:]) == 0
    assert torch.count_nonzero(g[3 * 64 * cfg["num_heads"][0]:, :]) == 0
    assert m.blocks[cfg["layer_num"]].fc1.weight.grad is None if cfg["layer_num"] < len(m.blocks) else True


def test_product_attention_on_host_matches_reference():
    fix = load_npz("autoformer_attention.npz")
    from cream_amd.autoformer import AttentionSuper
    att = AttentionSuper(256, num_heads=4, qkv_bias=True, relative_position=True, change_qkv=True)
    fill_params(att, seed=11)
    att.set_sample_config(sample_q_embed_dim=192, sample_num_heads=3, sample_in_embed_dim=216)
    g = torch.Generator().manual_seed(21)
    x = torch.randn(2, 197, 216, generator=g, requires_grad=True)
    gy = torch.randn(2, 197, 216, generator=g)
    y = att(x)
    y.backward(gy)
    assert max_rel(y.detach(), fix["y"]) < 1e-5
    assert max_rel(x.grad, fix["dx"]) < 1e-5
    for k, v in fix.items():
        if k.startswith("full|"):
            assert max_rel(dict(att.named_parameters())[k[5:]].grad, v) < 2e-5, k


def test_dense_forward_of_relative_position_module_still_available():
    from cream_amd.autoformer import RelativePosition2D_super
    rp = RelativePosition2D_super(64, 14)
    rp.set_sample_config(64)
    emb = rp(197, 197)
    want, _, _ = AO.rel_pos_embeddings(rp.embeddings_table_v.detach(), rp.embeddings_table_h.detach(), 197, 14)
    assert torch.equal(emb.detach(), want)


def test_evaluate_matches_direct_computation():
    """engine.evaluate (supernet_engine.py:113-160): one sub-network, loss / top-1 / top-5 averaged over
    samples, accumulated without per-batch host syncs."""
    import random
    import torch
    import torch.nn.functional as F
    from cream_amd.autoformer import engine
    torch.manual_seed(0)
    m = engine.build_supernet("T", drop_path_rate=0.1, depth=2, num_classes=10)
    ch = dict(engine.SEARCH_SPACES["T"]["choices"], depth=[1, 2])
    g = torch.Generator().manual_seed(1)
    batches = [(torch.randn(3, 3, 224, 224, generator=g), torch.randint(0, 10, (3,), generator=g)),
               (torch.randn(2, 3, 224, 224, generator=g), torch.randint(0, 10, (2,), generator=g))]
    random.seed(7)
    res = engine.evaluate(batches, m, amp_dtype=torch.float32, choices=ch, mode="super")
    random.seed(7)
    cfg = engine.sample_configs(ch)
    assert res["config"] == cfg and res["params"] == m.get_sampled_params_numel(cfg)
    m.eval()
    m.set_sample_config(cfg)
    with torch.no_grad():
        outs = torch.cat([m(x) for x, _ in batches])
    labels = torch.cat([y for _, y in batches])
    assert abs(res["loss"] - float(F.cross_entropy(outs, labels))) < 1e-5
    top5 = outs.topk(5, dim=1).indices
    assert abs(res["acc1"] - 100.0 * float((top5[:, 0] == labels).double().mean())) < 1e-9
    assert abs(res["acc5"] - 100.0 * float(top5.eq(labels[:, None]).any(1).double().mean())) < 1e-9
    res2 = engine.evaluate(batches, m, amp_dtype=torch.float32, mode="retrain", retrain_config=cfg)
    assert res2["loss"] == res["loss"]


def test_checkpoint_layout_and_resume(tmp_path):
    """On-disk format of supernet_train.py:363-370 and the resume rule of :316-330; the 'model' entry
    carries the reference's parameter names (golden key list of the reference's own state_dict)."""
    import json
    import torch
    from cream_amd.autoformer import engine
    torch.manual_seed(0)
    m = engine.build_supernet("T", depth=2)
    opt = engine.build_optimizer(m, batch_size=4)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=10)
    for p in m.parameters():                                   # one fake step so that the optimizer has state
        p.grad = torch.zeros_like(p)
    opt.step()
    sched.step()
    path = engine.save_checkpoint(str(tmp_path / "checkpoint.pth"), m, opt, sched, epoch=4, args={"model": "T"})
    assert engine.save_checkpoint(str(tmp_path / "other.pth"), m, rank=1) is None       # save_on_master
    ck = torch.load(path, map_location="cpu", weights_only=False)
    assert set(ck) == {"model", "optimizer", "lr_scheduler", "epoch", "args"} and ck["epoch"] == 4
    # the full supernet's checkpoint keys are the reference's own (golden list made by importing it)
    kat = json.load(open(os.path.join(ROOT, "tests", "golden", "autoformer_kat.json")))
    full = engine.build_supernet("T")
    assert sorted(full.state_dict().keys()) == kat["supernet_T_state_keys"]
    assert "blocks.0.attn.rel_pos_embed_k.embeddings_table_v" in ck["model"] and "blocks.1.fc1.weight" in ck["model"]
    m2 = engine.build_supernet("T", depth=2)
    opt2 = engine.build_optimizer(m2, batch_size=4)
    sched2 = torch.optim.lr_scheduler.CosineAnnealingLR(opt2, T_max=10)
    assert engine.load_checkpoint(path, m2, opt2, sched2) == 5
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    assert opt2.state_dict()["state"][0]["step"] == opt.state_dict()["state"][0]["step"]
    assert sched2.last_epoch == sched.last_epoch
    # a published weights-only file ({'model': ...}) or --eval: no training state is touched
    m3 = engine.build_supernet("T", depth=2)
    assert engine.load_checkpoint({"model": ck["model"]}, m3) == 0
    assert engine.load_checkpoint(path, m3, eval_only=True) == 0


def test_mixup_cutmix_batch_mode_properties():
    """Mixup / CutMix of the step (supernet_train.py:245-251 -> timm.data.Mixup, third-party and not vendored: restated,
    parity unpinned): soft targets sum to 1 and equal lam * onehot_s(y) + (1 - lam) * onehot_s(y flipped); mixup blends the
    batch with its flip; cutmix pastes exactly one box and lam is the surviving area; seeded numpy draws are reproducible."""
    import numpy as np
    from cream_amd.autoformer.data import Mixup
    fn = Mixup(num_classes=10)
    seen = set()
    for seed in range(12):
        np.random.seed(seed)
        x0 = torch.randn(4, 3, 32, 32)
        y = torch.tensor([1, 3, 5, 7])
        np.random.seed(seed)
        x, t = fn(x0.clone(), y)
        np.random.seed(seed)
        lam, cut = fn._params_per_batch()
        seen.add(cut)
        assert torch.allclose(t.sum(1), torch.ones(4), atol=1e-6)
        if cut:
            yl, yh, xl, xh = fn.rand_bbox(32, 32, lam)
            lam = 1.0 - (yh - yl) * (xh - xl) / 1024.0
            ref = x0.clone()
            ref[:, :, yl:yh, xl:xh] = x0.flip(0)[:, :, yl:yh, xl:xh]
            assert torch.equal(x, ref)
        else:
            assert torch.allclose(x, x0 * lam + x0.flip(0) * (1 - lam), atol=1e-6)
        on, off = 0.9 + 0.01, 0.01
        oh = torch.full((4, 10), off).scatter_(1, y.view(-1, 1), on)
        assert torch.allclose(t, oh * lam + oh.flip(0) * (1 - lam), atol=1e-6)
    assert seen == {True, False}

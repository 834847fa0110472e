"""GPU parity of the AutoFormer supernet step against golden vectors produced by the
reference itself (tests/golden, fp32 CPU): logits and gradients within 1e-3 relative in
fp32 mode (BASELINE.json's bar); the bf16 throughput mode is held to a looser, documented
tolerance (PyTorch's own CPU bf16 autocast of this model is 4e-3..9e-3 off)."""
import os
import sys

import pytest
import torch

from conftest import ROOT
from helpers import check_against_fixture, config_of, load_npz, max_rel

sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from fixture_utils import fill_params, make_batch, model_kwargs  # noqa: E402

pytestmark = pytest.mark.gpu
IMPLS = ["bucketed", "fused"]


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _set_impl(model, impl):
    from cream_amd.autoformer import fused_attention
    if impl == "fused" and not fused_attention.available():
        pytest.skip("fused attention kernels not built")
    for m in model.modules():
        if hasattr(m, "attention_impl"):
            m.attention_impl = impl


def _step(size, batch, impl, amp):
    from cream_amd.autoformer import Vision_TransformerSuper
    from cream_amd.autoformer.engine import soft_target_cross_entropy
    fix = load_npz(f"autoformer_{size}_step.npz")
    cfg = config_of(fix)
    m = Vision_TransformerSuper(**model_kwargs(size))
    fill_params(m, seed=3)
    m = m.to(_dev())
    _set_impl(m, impl)
    m.set_sample_config(cfg)
    m.train()
    images, target = make_batch(batch, seed=5)
    images, target = images.to(_dev()), target.to(_dev())
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
        logits = m(images)
        loss = soft_target_cross_entropy(logits, target)
    loss.backward()
    grads = {k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in m.named_parameters()}
    return fix, cfg, m, logits, loss, grads


@pytest.mark.parametrize("impl", IMPLS)
@pytest.mark.parametrize("size,batch", [("T", 2), ("S", 1)])
def test_step_fp32_within_1e3_of_reference(size, batch, impl, monkeypatch):
    """BASELINE's bar on the framework's OWN kernels: every projection (qkv / proj / fc1 / fc2 / patch embedding / head),
    forward, dgrad and wgrad, runs on csrc/gemm_f32.hip (exact-fp32 matrix cores), every LayerNorm on the fp32 instantiation
    of csrc/block_ops.hip, attention on the fp32 instantiation of csrc/attn_rpe2d.hip — a vendor GEMM or the framework's
    layer_norm reaching a device tensor fails the test."""
    import torch.nn.functional as F
    from cream_amd.autoformer import modules, native_fp32

    def no_library(name, fn):
        def guard(x, *a, **k):
            assert not x.is_cuda, f"{name} reached a device tensor: the fp32 parity step must run on the own kernels"
            return fn(x, *a, **k)
        return guard
    monkeypatch.setattr(modules.F, "linear", no_library("F.linear", F.linear))
    monkeypatch.setattr(modules.F, "layer_norm", no_library("F.layer_norm", F.layer_norm))
    before = dict(native_fp32.CALLS)
    fix, cfg, m, logits, loss, grads = _step(size, batch, impl, amp=False)
    depth = cfg["layer_num"]
    assert native_fp32.CALLS["linear"] - before["linear"] == 4 * depth + 2          # + patch embedding + head
    assert native_fp32.CALLS["layer_norm"] - before["layer_norm"] == 2 * depth + 1
    worst = check_against_fixture(fix, logits, loss, grads, tol=1e-3)
    print(f"[fp32 step {size} B={batch} {impl} on the own fp32 kernels] worst rel err vs the reference-made fixture {worst:.2e}")
    assert worst < 1e-3
    E = cfg["embed_dim"][0]
    g = m.blocks[0].attn.qkv.weight.grad
    assert torch.count_nonzero(g[:, E:]) == 0                              # nothing outside the slice
    assert torch.count_nonzero(g[3 * 64 * cfg["num_heads"][0]:, :]) == 0
    for t in ("k", "v"):                                                    # table rows 1 and 29 unused
        tg = getattr(m.blocks[0].attn, f"rel_pos_embed_{t}").embeddings_table_v.grad
        assert torch.count_nonzero(tg[1]) == 0 and torch.count_nonzero(tg[29]) == 0


# bounds = 2x measured on the MI355X (profiles/r03_parity.txt: logits 5.7e-3, loss 2.6e-4, worst full gradient 8.2e-3 — a
# position table of the last block); round 2 held this test to 3e-2 / 1e-2 / 8e-2
BF16_T_TOL = dict(logits=1.2e-2, loss=6e-4, grads=1.7e-2)


@pytest.mark.parametrize("impl", IMPLS)
def test_step_bf16_autocast_documented_tolerance(impl):
    fix, cfg, m, logits, loss, grads = _step("T", 2, impl, amp=True)
    el = max_rel(logits.detach().float().cpu(), fix["logits"])
    eloss = abs(float(loss) - float(fix["loss"][0])) / float(fix["loss"][0])
    eg = {k[5:]: max_rel(grads[k[5:]].float().cpu(), v) for k, v in fix.items() if k.startswith("full|")}
    worst = max(eg, key=eg.get)
    print(f"[T step B=2 bf16 {impl}] logits {el:.2e} loss {eloss:.2e} worst full-gradient {eg[worst]:.2e} ({worst})")
    assert el < BF16_T_TOL["logits"] and eloss < BF16_T_TOL["loss"]
    for k, e in eg.items():
        assert e < BF16_T_TOL["grads"], (k, e)


@pytest.mark.parametrize("impl", IMPLS)
def test_attention_module_fp32(impl):
    from cream_amd.autoformer import AttentionSuper
    fix = load_npz("autoformer_attention.npz")
    att = AttentionSuper(256, num_heads=4, qkv_bias=True, relative_position=True, change_qkv=True)
    fill_params(att, seed=11)
    att = att.to(_dev())
    _set_impl(att, impl)
    att.set_sample_config(sample_q_embed_dim=192, sample_num_heads=3, sample_in_embed_dim=216)
    g = torch.Generator().manual_seed(21)
    x = torch.randn(2, 197, 216, generator=g).to(_dev()).requires_grad_()
    gy = torch.randn(2, 197, 216, generator=g).to(_dev())
    y = att(x)
    y.backward(gy)
    assert max_rel(y.detach().cpu(), fix["y"]) < 1e-3
    assert max_rel(x.grad.cpu(), fix["dx"]) < 1e-3
    for k, v in fix.items():
        if k.startswith("full|"):
            assert max_rel(dict(att.named_parameters())[k[5:]].grad.cpu(), v) < 1e-3, k


def test_trainer_step_runs_and_updates_only_with_finite_values():
    from cream_amd import comm
    from cream_amd.autoformer import engine
    torch.manual_seed(0)
    model = engine.build_supernet("T", drop_path_rate=0.1).to(_dev())
    opt = engine.build_optimizer(model, batch_size=8)
    tr = engine.SupernetTrainer(model, opt, engine.SEARCH_SPACES["T"]["choices"], comm.GradReducer(model))
    images = torch.randn(8, 3, 224, 224, device=_dev())
    target = torch.softmax(torch.randn(8, 1000, device=_dev()), -1)
    tr.start_epoch(0)
    before = model.head.weight.detach().clone()
    losses = [float(tr.step(images, target)) for _ in range(3)]
    assert all(map(lambda v: v == v and abs(v) < 1e4, losses))
    assert not torch.equal(before, model.head.weight.detach())
    # golden draw sequence of epoch 0: third config drawn last
    assert tr.config["layer_num"] in (12, 13, 14)


def test_full_size_step_is_bit_reproducible_and_respects_sampled_slices():
    """BASELINE size (supernet-S, B = 128, 224^2, bf16 throughput mode, native block sequencing with
    the weight-gradient GEMMs + gradient finalisation on the side stream): two identical steps give
    identical bits (a race between the two streams, a recycled buffer or an atomics-ordered sum
    would show up here), gradients are exactly zero outside the sampled slices of the super
    weights, and blocks beyond the sampled depth receive none."""
    from cream_amd.autoformer import engine
    dev = _dev()
    torch.manual_seed(0)
    m = engine.build_supernet("S", drop_path_rate=0.1).to(dev)
    cfg = dict(layer_num=13, embed_dim=[384] * 13, num_heads=[6, 5, 7, 6, 6, 5, 7, 7, 5, 6, 6, 7, 5],
               mlp_ratio=[3.5, 3.0, 4.0, 3.5, 3.0, 4.0, 3.5, 3.5, 3.0, 4.0, 4.0, 3.0, 3.5])
    m.set_sample_config(cfg)
    m.train()
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.randn(128, 3, 224, 224, device=dev, generator=g)
    t = torch.softmax(torch.randn(128, 1000, device=dev, generator=g), -1)
    runs = []
    for _ in range(2):
        m.zero_grad(set_to_none=False)
        torch.manual_seed(123)                               # same drop-path draws
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = engine.soft_target_cross_entropy(m(x), t)
        loss.backward()
        torch.cuda.synchronize()
        runs.append((float(loss), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}))
    assert runs[0][0] == runs[1][0] and runs[0][0] == runs[0][0]
    for k, v in runs[0][1].items():
        assert torch.equal(v, runs[1][1][k]), k
    gr = runs[0][1]
    assert torch.isfinite(gr["blocks.0.fc1.weight"]).all()
    assert torch.count_nonzero(gr["blocks.0.fc1.weight"][int(384 * 3.5):]) == 0      # rows beyond F
    assert torch.count_nonzero(gr["blocks.0.fc1.weight"][:, 384:]) == 0               # columns beyond E
    assert torch.count_nonzero(gr["blocks.1.attn.qkv.weight"][3 * 320:]) == 0         # rows beyond 3Q (H = 5)
    assert torch.count_nonzero(gr["blocks.1.attn.qkv.weight"][:3 * 320, :384]) > 0
    assert torch.count_nonzero(gr["blocks.2.attn.proj.weight"][:384, :448]) > 0
    assert "blocks.13.fc1.weight" not in gr or torch.count_nonzero(gr["blocks.13.fc1.weight"]) == 0


def test_join_at_the_end_of_the_backward_pass_gives_the_same_gradients(monkeypatch):
    """block.join_side_stream_at_end_of_backward: the main stream waits for the weight-gradient stream in autograd's end-of-pass
    callback (the stem's backward then runs under the side stream's tail) instead of right behind the blocks.  Same step both ways
    at the benchmark size, read back right after backward() returns: every gradient bit-identical (a missing or late join, or a
    workspace recycled under the side stream, shows up as a difference), three times in a row."""
    from cream_amd.autoformer import block as K
    from cream_amd.autoformer import engine
    dev = _dev()
    torch.manual_seed(0)
    m = engine.build_supernet("S", drop_path_rate=0.1).to(dev)
    m.set_sample_config(dict(layer_num=13, embed_dim=[384] * 13, num_heads=[6] * 13, mlp_ratio=[3.5] * 13))
    m.train()
    g = torch.Generator(device=dev).manual_seed(7)
    x = torch.randn(128, 3, 224, 224, device=dev, generator=g)
    t = torch.softmax(torch.randn(128, 1000, device=dev, generator=g), -1)

    def grads(defer):
        monkeypatch.setattr(K, "DEFER_JOIN", defer)
        m.zero_grad(set_to_none=False)
        torch.manual_seed(123)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = engine.soft_target_cross_entropy(m(x), t)
        loss.backward()
        # no synchronize: what the optimizer would read on this stream right after backward()
        return {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}

    ref = grads(False)
    for _ in range(3):
        got = grads(True)
        for k, v in ref.items():
            assert torch.equal(v, got[k]), k


def test_subnet_evaluation_native_path_matches_module_path():
    """engine.evaluate (supernet_engine.py:113-160) in bf16 on the GPU: the run of blocks goes through
    the native forward sequencing (eval mode, no drop-path, no autograd graph); same sub-network and
    batches through the module-by-module path must agree to bf16 accuracy."""
    import random
    from cream_amd.autoformer import engine
    dev = _dev()
    torch.manual_seed(0)
    m = engine.build_supernet("S", drop_path_rate=0.1, num_classes=100).to(dev)
    g = torch.Generator(device=dev).manual_seed(3)
    batches = [(torch.randn(8, 3, 224, 224, device=dev, generator=g), torch.randint(0, 100, (8,), device=dev, generator=g))
               for _ in range(2)]
    ch = engine.SEARCH_SPACES["S"]["choices"]
    res = {}
    for fused in (True, False):
        for blk in m.blocks:
            blk.fused = fused
        random.seed(11)
        res[fused] = engine.evaluate(batches, m, choices=ch, mode="super")
    assert res[True]["config"] == res[False]["config"] and res[True]["params"] == res[False]["params"]
    assert abs(res[True]["loss"] - res[False]["loss"]) < 2e-2 * abs(res[False]["loss"])
    assert res[True]["loss"] == res[True]["loss"]


# measured on the MI355X (round 3, profiles/r03_parity.txt): whole AutoFormer-S step, depth 13, E = 384, B = 128, native
# bf16 throughput path against the fp32 CPU oracle: logits 7.5e-3, loss 4.2e-5, projection / embedding weights 8.1e-3
# (worst: head.weight), small tensors 1.03e-2 (worst: blocks.1.ffn_layer_norm.weight) — bounds = 2x measured (loss: 2.5x).
STEP128_TOL = dict(logits=1.5e-2, loss=1e-4, weights=1.7e-2, small=2.1e-2)


# three sub-networks that together cover every embed dim and depth of supernet-S (the whole-step check at
# bench size covered one) — heads and MLP ratios mixed within each
STEP128_CONFIGS = {
    "d13_E384": dict(layer_num=13, embed_dim=[384] * 13, num_heads=[6, 5, 7, 6, 6, 5, 7, 7, 5, 6, 6, 7, 5],
                     mlp_ratio=[3.5, 3.0, 4.0, 3.5, 3.0, 4.0, 3.5, 3.5, 3.0, 4.0, 4.0, 3.0, 3.5]),
    "d12_E320": dict(layer_num=12, embed_dim=[320] * 12, num_heads=[5, 7, 6, 5, 6, 7, 7, 5, 6, 6, 5, 7],
                     mlp_ratio=[3.0, 4.0, 3.5, 3.5, 3.0, 4.0, 4.0, 3.0, 3.5, 3.0, 4.0, 3.5]),
    "d14_E448": dict(layer_num=14, embed_dim=[448] * 14, num_heads=[7, 6, 5, 7, 5, 6, 6, 7, 5, 5, 7, 6, 6, 7],
                     mlp_ratio=[4.0, 3.5, 3.0, 3.0, 4.0, 3.5, 3.5, 4.0, 3.0, 3.5, 3.0, 4.0, 3.5, 4.0]),
}


@pytest.mark.parametrize("name", list(STEP128_CONFIGS))
def test_whole_step_at_bench_size_matches_oracle(name):
    """The benchmarked workload itself: AutoFormer-S supernet, one sampled sub-network (depth 12 / 13 / 14 at E = 320 / 384 /
    448, mixed heads and MLP ratios), B = 128, 224^2 — the native bf16 path (block stack = one autograd node, every block one native call
    per direction, weight-gradient GEMMs on the side stream, native stem / tail) against ONE step of
    oracle.autoformer_oracle.train_step (the reference's dense fp32 formulation on the CPU): logits, loss and EVERY
    parameter gradient (max-abs error / max-abs reference per tensor; exact zeros outside the sampled slices)."""
    from cream_amd.autoformer import engine
    from oracle import autoformer_oracle as AO
    dev = _dev()
    m = engine.build_supernet("S", drop_path_rate=0.0)
    fill_params(m, seed=7)
    cfg = STEP128_CONFIGS[name]
    E0, depth = cfg["embed_dim"][0], cfg["layer_num"]
    images, target = make_batch(128, seed=9)
    sd = {k: v.detach().clone() for k, v in m.named_parameters()}
    torch.set_num_threads(min(64, os.cpu_count() or 1))
    loss_ref, grads_ref = AO.train_step(sd, cfg, images, target)
    with torch.no_grad():
        logits_ref = AO.forward(sd, cfg, images)
    m = m.to(dev)
    m.set_sample_config(cfg)
    m.train()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        logits = m(images.to(dev))
        loss = engine.soft_target_cross_entropy(logits, target.to(dev))
    loss.backward()
    torch.cuda.synchronize()
    errs = dict(logits=max_rel(logits.detach().float().cpu(), logits_ref),
                loss=abs(float(loss) - float(loss_ref)) / abs(float(loss_ref)))
    worst = dict(weights=(0.0, ""), small=(0.0, ""))
    for k, p in m.named_parameters():
        ref = grads_ref[k]
        got = (p.grad if p.grad is not None else torch.zeros_like(p)).detach().float().cpu()
        if float(ref.abs().max()) == 0.0:                         # blocks beyond the sampled depth
            assert float(got.abs().max()) == 0.0, k
            continue
        e = max_rel(got, ref)
        cls = "weights" if p.dim() == 2 and min(p.shape) >= 64 else "small"
        if e > worst[cls][0]:
            worst[cls] = (e, k)
    errs["weights"], errs["small"] = worst["weights"][0], worst["small"][0]
    print(f"[whole step S {name} B128 bf16 native vs fp32 oracle]", {k: f"{v:.2e}" for k, v in errs.items()},
          "worst tensors:", worst["weights"][1], "/", worst["small"][1])
    i5 = cfg["num_heads"].index(5)
    g = m.blocks[i5].attn.qkv.weight.grad                         # H = 5: rows beyond 3 * 320 and columns beyond E stay zero
    assert torch.count_nonzero(g[3 * 320:]) == 0
    assert E0 == g.shape[1] or torch.count_nonzero(g[:, E0:]) == 0
    if depth < len(m.blocks):
        gb = m.blocks[depth].attn.qkv.weight.grad
        assert gb is None or torch.count_nonzero(gb) == 0
    for k, tol in STEP128_TOL.items():
        assert errs[k] < tol, (k, errs[k], worst)


def test_checkpoint_resume_on_the_device_continues_bit_for_bit(tmp_path):
    """SURVEY 8f-4 on the device: the dictionary of supernet_train.py:363-370 written from a model that has trained on the native path
    (one-launch AdamW with its bf16 operand copies and table images), loaded into a FRESH model + optimizer by the resume rule of
    :316-330 — the operand copies are derived data, re-derived on load — and both take the same next steps: identical losses and
    identical weights, bit for bit.  The device Mixup / CutMix launch runs in the step body (numpy-seeded identically)."""
    import random
    import numpy as np
    from cream_amd import comm
    from cream_amd.autoformer import engine
    from cream_amd.autoformer.data import Mixup

    def make():
        torch.manual_seed(0)
        model = engine.build_supernet("T", drop_path_rate=0.0).to(_dev())
        opt = engine.build_optimizer(model, batch_size=8)
        tr = engine.SupernetTrainer(model, opt, engine.SEARCH_SPACES["T"]["choices"], comm.GradReducer(model), mixup_fn=Mixup())
        return model, opt, tr

    g = torch.Generator(device=_dev()).manual_seed(5)
    images = torch.randn(8, 3, 224, 224, device=_dev(), generator=g)
    labels = torch.randint(0, 1000, (8,), device=_dev(), generator=g)
    model, opt, tr = make()
    tr.start_epoch(0)
    np.random.seed(0)
    for _ in range(2):
        tr.step(images.clone(), labels)
    path = engine.save_checkpoint(str(tmp_path / "checkpoint.pth"), model, opt, None, epoch=0, args={"model": "T"})
    ck = torch.load(path, map_location="cpu", weights_only=False)
    assert "blocks.0.attn.rel_pos_embed_k.embeddings_table_v" in ck["model"] and not any("bf16" in k or "timg" in k for k in ck["model"])
    rng_py, rng_np = random.getstate(), np.random.get_state()
    more = [float(tr.step(images.clone(), labels)) for _ in range(2)]

    model2, opt2, tr2 = make()
    with torch.no_grad():                                        # a fresh model with OTHER weights: everything must come from the file
        for p in model2.parameters():
            p.add_(0.01)
    ck2 = dict(ck)
    ck2.setdefault("lr_scheduler", {})                           # (the resume rule wants all three entries)
    assert engine.load_checkpoint(ck2, model2, opt2, None) == 1
    tr2.start_epoch(0)
    random.setstate(rng_py)
    np.random.set_state(rng_np)
    again = [float(tr2.step(images.clone(), labels)) for _ in range(2)]
    assert again == more, (again, more)
    for (k, a), (_, b) in zip(model.state_dict().items(), model2.state_dict().items()):
        assert torch.equal(a, b), k

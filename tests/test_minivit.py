"""Mini-DeiT (cream_amd/minivit.py) against fixtures made by running the reference's own model classes
(MiniViT/Mini-DeiT/mini_vision_transformer.py, tests/golden/make_golden.py `minivit`): state-dict keys, parameter count,
logits and the gradient of every parameter.  CPU: fp32.  GPU: fp32 on the HIP rpe_index operator, and bf16 autocast,
where the configuration without head transforms must take the fused iRPE attention kernels."""
import sys
import os
from functools import partial

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from fixture_utils import grad_digest  # noqa: E402,F401
from helpers import load_json, load_npz, max_rel  # noqa: E402
from make_golden import MINIVIT_CASES, minivit_fill, zlib_seed  # noqa: E402


def build(tag):
    from cream_amd import minivit
    from cream_amd.irpe import get_rpe_config
    c = MINIVIT_CASES[tag]
    torch.manual_seed(0)
    if c['registered']:
        model = minivit.mini_deit('tiny')
    else:
        cfg = get_rpe_config(ratio=1.9, method='product', mode='ctx', shared_head=True, skip=c['skip'], rpe_on=c['rpe_on'])
        model = minivit.MiniVisionTransformer(patch_size=16, embed_dim=192, depth=c['depth'], num_heads=3, mlp_ratio=4,
                                              qkv_bias=True, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), rpe_config=cfg,
                                              use_cls_token=c['use_cls_token'], repeated_times=c['repeated_times'],
                                              use_transform=c['use_transform'], drop_path_rate=c['drop_path_rate'])
    minivit_fill(model, seed=23)
    return model.eval()


def run(model, tag, device, autocast=False):
    g = torch.Generator().manual_seed(zlib_seed(tag))
    x = torch.randn(2, 3, 224, 224, generator=g).to(device)
    gy = torch.randn(2, 1000, generator=g).to(device)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
        logits = model(x)
    (logits.float() * gy).sum().backward()
    return logits, {k: p.grad for k, p in model.named_parameters()}


def compare(tag, logits, grads, tol):
    fix = load_npz("minivit.npz")
    errs = {"logits": max_rel(logits.detach().cpu().float(), fix[f"{tag}|logits"])}
    for k, v in fix.items():
        if k.startswith(tag + "|") and k.endswith("|norm"):
            name = k[len(tag) + 1:-5]
            ref = float(v[0])
            g = grads[name].detach().cpu().double().flatten()
            if ref < 1e-5:
                # exactly zero in exact arithmetic (last layer, class-token row only: every key of that row is in the skip
                # bucket, and a softmax row's score gradients sum to zero) — the reference's value is rounding noise
                assert float(g.norm()) < tol, (name, float(g.norm()))
                continue
            errs[name + "|norm"] = abs(float(g.norm()) - ref) / ref
            scale = ref / max(1.0, g.numel()) ** 0.5
            sample = torch.from_numpy(fix[f"{tag}|{name}|sample"])
            errs[name + "|sample"] = float((g[::997] - sample).abs().max() / scale) / 10.0
    bad = {k: e for k, e in errs.items() if not e <= tol}
    assert not bad, f"{tag}: exceeds {tol}: " + ", ".join(f"{k}={e:.2e}" for k, e in sorted(bad.items(), key=lambda t: -t[1])[:8])
    return max(errs.values())


@pytest.mark.parametrize("tag", list(MINIVIT_CASES))
def test_minivit_matches_reference_on_cpu(tag):
    model = build(tag)
    meta = load_json("minivit.json")[tag]
    assert list(model.state_dict().keys()) == meta["keys"]
    assert sum(p.numel() for p in model.parameters()) == meta["n_params"]
    logits, grads = run(model, tag, "cpu")
    worst = compare(tag, logits, grads, 1e-4)
    print(f"[minivit cpu {tag}] worst {worst:.2e}")


def test_repeat_counter_selects_the_instances():
    """Each repeat must see its own tables / norms / transforms: zeroing repeat 1's iRPE table changes the output, and
    it changes nothing when the block is only run for repeat 0."""
    from cream_amd import minivit
    from cream_amd.irpe import get_rpe_config
    cfg = get_rpe_config(ratio=1.9, method='product', mode='ctx', shared_head=True, skip=0, rpe_on='k')
    torch.manual_seed(1)
    blk = minivit.RepeatedMiniBlock(repeated_times=2, dim=128, num_heads=2, qkv_bias=True, rpe_config=cfg, drop_paths=[0., 0.],
                                    use_transform=True).eval()
    with torch.no_grad():
        for m in blk.block.attn.rpe_k.instances:
            m.lookup_table_weight.normal_(std=0.3)
    x = torch.randn(1, 16, 128)
    with torch.no_grad():
        y = blk(x)
        blk._set_repeat(0)
        y0 = blk.block(x)
        blk.block.attn.rpe_k.instances[1].lookup_table_weight.zero_()
        assert not torch.equal(blk(x), y)
        blk._set_repeat(0)
        assert torch.equal(blk.block(x), y0)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", list(MINIVIT_CASES))
def test_minivit_matches_reference_on_gpu_fp32(tag):
    from cream_amd import timing
    model = build(tag).to("cuda:0")
    timing.reset()
    timing.enable(True)
    logits, grads = run(model, tag, "cuda:0")
    timing.enable(False)
    names = set(timing.summary())
    assert "rpe_index_fwd" in names and "rpe_index_bwd" in names, names      # the HIP operator ran, not an eager gather
    worst = compare(tag, logits, grads, 1e-3)
    print(f"[minivit gpu fp32 {tag}] worst {worst:.2e}")


@pytest.mark.gpu
def test_minivit_without_head_transforms_takes_the_fused_kernels():
    from cream_amd import timing
    tag = "shared_qkv_cls"
    model = build(tag).to("cuda:0")
    timing.reset()
    timing.enable(True)
    logits, grads = run(model, tag, "cuda:0", autocast=True)
    timing.enable(False)
    names = set(timing.summary())
    assert {"irpe_attn_fwd", "irpe_attn_bwd"} <= names and not {"rpe_index_fwd", "rpe_index_bwd"} & names, names
    worst = compare(tag, logits, grads, 4e-2)         # bf16 operands end to end (the reference's own autocast: ~1e-2)
    print(f"[minivit gpu bf16 fused {tag}] worst {worst:.2e}")

"""DeiT-with-iRPE block stack on the own kernels (cream_amd/deit_native.py: one autograd node for all RPEBlocks under bf16 autocast)
against the same model evaluated in fp32 without autocast (rpe_vision_transformer.py:100-117, :193-199) — logits and EVERY parameter
gradient, the lookup tables of the rpe terms included — next to the module path under the same autocast (fused attention, framework
linears / LayerNorm / GELU), whose distance to fp32 is the yardstick."""
import copy
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def _run_blocks(blocks, x, g, autocast, native):
    """All blocks on a (B, L, D) input with an upstream gradient on EVERY token (the classifier alone reaches the blocks through the
    class token only: the last block's table gradients would be pure rounding noise)."""
    for p in blocks.parameters():
        p.grad = None
    x = x.clone().requires_grad_()
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
        if native:
            from cream_amd import deit_native
            assert deit_native.supported(blocks, x)
            out = deit_native.run(blocks, x)
        else:
            out = x
            for blk in blocks:
                out = blk(out)
    out.float().backward(g)
    torch.cuda.synchronize()
    return out.float().detach().clone(), x.grad.clone(), {n: p.grad.detach().clone() for n, p in blocks.named_parameters()}


@pytest.mark.parametrize("rpe_on,mode,size,shared", [("qkv", "ctx", "tiny", True), ("k", "ctx", "small", False), ("qk", "bias", "tiny", False),
                                                     ("", "ctx", "tiny", True)])
def test_native_block_stack_matches_fp32_blocks(rpe_on, mode, size, shared):
    from cream_amd import timing
    from cream_amd.rpe_attention import deit_irpe
    torch.manual_seed(7)
    model = deit_irpe(size, rpe_on=rpe_on or "k", mode=mode, shared_head=shared, depth=3, num_classes=10).to(DEV)
    if not rpe_on:
        for blk in model.blocks:
            blk.attn.rpe_k = None                               # plain DeiT blocks through the same node
    with torch.no_grad():                                       # the zoo initialises the tables to zero
        for n, p in model.named_parameters():
            if "lookup_table" in n:
                p.normal_(0, 0.2)
            elif n.endswith("bias"):
                p.normal_(0, 0.05)
            elif "norm" in n and n.endswith("weight"):
                p.add_(0.1 * torch.randn_like(p))
    B, L, D = 4, 197, model.embed_dim
    x = torch.randn(B, L, D, device=DEV)
    g = torch.randn(B, L, D, device=DEV)
    ref = _run_blocks(copy.deepcopy(model.blocks), x, g, autocast=False, native=False)
    mod = _run_blocks(copy.deepcopy(model.blocks), x, g, autocast=True, native=False)
    timing.reset(); timing.enable(True)
    os.environ["CREAM_DEIT_NATIVE"] = "1"
    nat = _run_blocks(copy.deepcopy(model.blocks), x, g, autocast=True, native=True)
    timing.enable(False)
    assert {"irpe_attn_fwd", "irpe_attn_bwd"} <= set(timing.summary())
    errs = {"out": (_rel(nat[0], ref[0]), _rel(mod[0], ref[0])), "dx": (_rel(nat[1], ref[1]), _rel(mod[1], ref[1]))}
    for n in ref[2]:
        assert nat[2][n] is not None and nat[2][n].shape == ref[2][n].shape and torch.isfinite(nat[2][n]).all(), n
        errs[n] = (_rel(nat[2][n], ref[2][n]), _rel(mod[2][n], ref[2][n]))
    top = sorted(errs.items(), key=lambda kv: -kv[1][0])[:4]
    print(f"[deit native {size} {rpe_on or 'none'} {mode}] out / dx native {errs['out'][0]:.2e} / {errs['dx'][0]:.2e}, module "
          f"{errs['out'][1]:.2e} / {errs['dx'][1]:.2e}; worst (native, module):", [(n, f"{a:.2e}", f"{b:.2e}") for n, (a, b) in top])
    for n, (a, b) in errs.items():
        # bf16 GEMM operands with fp32 accumulation and an fp32 residual stream: within 3e-2 of the fp32 blocks, or no worse than twice
        # the module path under the same autocast (the bucket gradients of a near-empty bucket are rounding noise in both)
        assert a < max(3e-2, 2 * b), (n, a, b)


def test_native_stack_with_stochastic_depth_matches_fp32_blocks_under_the_same_masks():
    """DropPath (rpe_vision_transformer.py:115-116: x + drop_path(branch(x)); timm's per-sample mask / keep) inside the node: the
    per-sample factors go through the residual / LayerNorm kernels.  Same factors in an fp32 evaluation of the blocks; a dropped
    sample's branch must contribute neither to the output nor to any gradient."""
    from cream_amd import deit_native
    from cream_amd.rpe_attention import deit_irpe
    from cream_amd.tinyclip import native
    torch.manual_seed(11)
    model = deit_irpe("tiny", rpe_on="k", depth=3, num_classes=10, drop_path_rate=0.3).to(DEV).train()
    with torch.no_grad():
        for n, p in model.named_parameters():
            if "lookup_table" in n:
                p.normal_(0, 0.2)
    B, L, D = 4, 197, model.embed_dim
    x = torch.randn(B, L, D, device=DEV)
    g = torch.randn(B, L, D, device=DEV)
    gen = torch.Generator(device=DEV).manual_seed(5)
    fixed = []
    for blk in model.blocks:
        keep = 1.0 - float(getattr(blk.drop_path, "drop_prob", 0.0) or 0.0)
        fixed.append(tuple(torch.floor(keep + torch.rand(B, device=DEV, generator=gen)) / keep for _ in range(2)))
    fixed[1] = (torch.tensor([0.0, 1 / 0.85, 0.0, 1 / 0.85], device=DEV), fixed[1][1])          # dropped samples for certain
    ref_blocks = copy.deepcopy(model.blocks)
    xr = x.clone().requires_grad_()
    out = xr
    for blk, (sa, sm) in zip(ref_blocks, fixed):
        out = out + sa[:, None, None] * blk.attn(blk.norm1(out))
        out = out + sm[:, None, None] * blk.mlp(blk.norm2(out))
    out.backward(g)
    it = iter(fixed)
    orig = native._path_scales
    native._path_scales = lambda view, B_, dev, training: next(it)
    try:
        nat_blocks = copy.deepcopy(model.blocks)
        xn = x.clone().requires_grad_()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            assert deit_native.supported(nat_blocks, xn)
            o = deit_native.run(nat_blocks, xn)
        o.float().backward(g)
    finally:
        native._path_scales = orig
    errs = {"out": _rel(o.float(), out), "dx": _rel(xn.grad, xr.grad)}
    for (n, p), (_, q) in zip(nat_blocks.named_parameters(), ref_blocks.named_parameters()):
        errs[n] = _rel(p.grad, q.grad)
    print("[deit native drop_path]", {k: f"{v:.2e}" for k, v in sorted(errs.items(), key=lambda kv: -kv[1])[:5]})
    assert all(v < 3e-2 for v in errs.values()), errs
    # the real draw: masks from the device generator, a different one per call
    with torch.autocast("cuda", dtype=torch.bfloat16):
        a, b = deit_native.run(model.blocks, x), deit_native.run(model.blocks, x)
    assert not torch.equal(a, b)
    model.eval()
    with torch.autocast("cuda", dtype=torch.bfloat16), torch.no_grad():
        assert torch.equal(deit_native.run(model.blocks, x), deit_native.run(model.blocks, x))


def test_whole_model_takes_the_native_stack_under_autocast():
    """forward_features (rpe_vision_transformer.py:193-199) dispatches to the node; logits against the fp32 model, every gradient set."""
    from cream_amd import timing
    from cream_amd.rpe_attention import deit_irpe
    torch.manual_seed(3)
    model = deit_irpe("tiny", rpe_on="k", depth=2, num_classes=100).to(DEV)
    x = torch.randn(4, 3, 224, 224, device=DEV)
    y = torch.randint(0, 100, (4,), device=DEV)
    with torch.no_grad():
        ref = model(x)
    calls = []
    from cream_amd.tinyclip import native
    orig = native.stack
    native.stack = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            logits = model(x)
        torch.nn.functional.cross_entropy(logits.float(), y).backward()
    finally:
        native.stack = orig
    assert calls == [1]
    assert _rel(logits.float(), ref) < 2e-2
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())


def test_native_stack_is_not_taken_with_projection_dropout_frozen_parameters_or_the_cross_method():
    from cream_amd import deit_native
    from cream_amd.rpe_attention import deit_irpe
    x = torch.empty(2, 197, 192, device=DEV)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        m = deit_irpe("tiny", rpe_on="k", depth=2, drop_path_rate=0.1, drop_rate=0.1).to(DEV)
        assert not deit_native.supported(m.blocks, x)            # training mode: projection / MLP dropout is active
        m.eval()
        assert deit_native.supported(m.blocks, x)
        m = deit_irpe("tiny", rpe_on="k", depth=2).to(DEV)
        m.blocks[1].mlp.fc1.weight.requires_grad_(False)
        assert not deit_native.supported(m.blocks, x)
        with torch.no_grad():
            assert deit_native.supported(m.blocks, x)
        m = deit_irpe("tiny", rpe_on="k", method="cross", depth=2).to(DEV)
        assert not deit_native.supported(m.blocks, x)            # cross: the one-table view goes through autograd (module path)
    m = deit_irpe("tiny", rpe_on="k", depth=2).to(DEV)
    assert not deit_native.supported(m.blocks, x)                # no autocast: fp32 parity route

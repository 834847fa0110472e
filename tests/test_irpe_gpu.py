"""iRPE on the MI355X: the modules of cream_amd.irpe / rpe_attention on the HIP rpe_index
gather/scatter, against the reference's golden vectors and against the CPU execution of the
same modules; BASELINE config 4 shapes (DeiT-base-384: H=12, L=577, 50 buckets)."""
import os
import sys

import pytest
import torch

from conftest import ROOT
from helpers import load_npz, max_rel

sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from fixture_utils import fill_params  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_rpe_attention_golden_on_gpu():
    from cream_amd import irpe as I
    from cream_amd.rpe_attention import RPEAttention
    fix = load_npz("irpe_attention.npz")
    cfg = I.get_rpe_config(ratio=1.9, method="product", mode="ctx", shared_head=True, skip=1, rpe_on="qkv")
    att = RPEAttention(192, num_heads=3, qkv_bias=True, rpe_config=cfg)
    fill_params(att, seed=19)
    with torch.no_grad():
        for n, p in att.named_parameters():
            if "lookup_table" in n:
                p.copy_(0.3 * torch.randn(p.shape, generator=torch.Generator().manual_seed(len(n))))
    att = att.to(DEV)
    g = torch.Generator().manual_seed(41)
    x = torch.randn(2, 197, 192, generator=g).to(DEV).requires_grad_()
    gy = torch.randn(2, 197, 192, generator=g).to(DEV)
    y = att(x)
    y.backward(gy)
    assert max_rel(y.detach().cpu(), fix["y"]) < 1e-3 and max_rel(x.grad.cpu(), fix["dx"]) < 1e-3
    for k, v in fix.items():
        if k.startswith("full|"):
            assert max_rel(dict(att.named_parameters())[k[5:]].grad.cpu(), v) < 1e-3, k


@pytest.mark.parametrize("shared", [True, False])
def test_config4_shapes_gpu_equals_cpu(shared):
    """DeiT-base-384 geometry (24x24 grid + class token), rpe on q, k and v."""
    from cream_amd import irpe as I
    cfg = I.get_rpe_config(ratio=1.9, method="product", mode="ctx", shared_head=shared, skip=1, rpe_on="qkv")
    mods = I.build_rpe(cfg, head_dim=64, num_heads=12)
    g = torch.Generator().manual_seed(7)
    B, H, L = 2, 12, 577
    for which, mod in zip("qkv", mods):
        with torch.no_grad():
            for p in mod.parameters():
                p.copy_(0.3 * torch.randn(p.shape, generator=g))
        x = (torch.randn(B, H, L, L, generator=g).softmax(-1) if which == "v" else torch.randn(B, H, L, 64, generator=g))
        gy = torch.randn(B, H, L, 64 if which == "v" else L, generator=g)
        res = []
        for dev in ("cpu", DEV):
            m = mod.to(dev)
            xx = x.to(dev).requires_grad_()
            y = m(xx)
            gr = torch.autograd.grad(y, [xx] + list(m.parameters()), gy.to(dev))
            res.append([y.detach().cpu()] + [t.cpu() for t in gr])
        mod.to("cpu")
        for a, b in zip(res[1], res[0]):
            assert max_rel(a, b) < 1e-4
        assert int(mod.bucket_ids(x).sum()) == 8019121                 # SURVEY §4 checksum


def test_deit_base_384_attention_bf16_runs_and_is_close():
    from cream_amd import irpe as I
    from cream_amd.rpe_attention import RPEAttention
    cfg = I.get_rpe_config(ratio=1.9, method="product", mode="ctx", shared_head=True, skip=1, rpe_on="k")
    att = RPEAttention(768, num_heads=12, qkv_bias=True, rpe_config=cfg)
    fill_params(att, seed=23)
    att = att.to(DEV)
    x = torch.randn(4, 577, 768, device=DEV)
    ref = att(x)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = att(x)
    assert y.dtype == torch.bfloat16 and max_rel(y.detach().float().cpu(), ref.detach().cpu()) < 3e-2

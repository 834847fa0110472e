"""DETR's encoder self-attention with iRPE (cream_amd/detr_attention.py) against a fixture made by running the reference's
own RPEMultiheadAttention (iRPE/DETR-with-iRPE/models/rpe_attention, tests/golden/make_golden.py `detr`): rectangular
feature maps, skip = 0, head_dim 32, key padding and additive masks, contextual and bias mode.  CPU fp32 here; the GPU
run goes through the HIP rpe_index operator."""
import json
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from helpers import load_npz, max_rel  # noqa: E402
from make_golden import DETR_CASES, detr_fill, detr_inputs  # noqa: E402


def run_case(tag, device):
    from cream_amd.detr_attention import RPEMultiheadAttention
    from cream_amd.irpe import get_rpe_config
    c = DETR_CASES[tag]
    fix = {k[len(tag) + 1:]: v for k, v in load_npz("detr_rpe_attention.npz").items() if k.startswith(tag + "|")}
    att = RPEMultiheadAttention(256, 8, dropout=0.0, rpe_config=get_rpe_config(**c['kw']))
    assert list(att.state_dict().keys()) == json.loads(bytes(fix["keys"]).decode())
    detr_fill(att, seed=31)
    att.to(device)
    src, pos, gy, pad, add = (t.to(device) if t is not None else None for t in detr_inputs(tag, c))
    src.requires_grad_()
    pos.requires_grad_()
    qk = src + pos
    out, wts = att(qk, qk, src, key_padding_mask=pad, attn_mask=add, hw=c['hw'])
    (out * gy).sum().backward()
    errs = {}
    for name, t in (("out", out), ("dsrc", src.grad), ("dpos", pos.grad)):
        errs[name] = max_rel(t[::6, :, ::2].detach().cpu(), fix[name])
        ref = float(fix[name + "|norm"][0])
        errs[name + "|norm"] = abs(float(t.detach().double().norm()) - ref) / ref
    errs["weights"] = max_rel(wts[:, ::5, ::3].detach().cpu(), fix["weights"])
    for n, p in att.named_parameters():
        g = p.grad if ("lookup" in n or p.dim() == 1) else p.grad[::5, ::3]
        errs["grad|" + n] = max_rel(g.detach().cpu(), fix["grad|" + n])
    if pad is not None:                      # padded keys receive no attention at all
        assert float(wts[pad[:, None, :].expand_as(wts)].abs().max()) == 0.0
    return errs


@pytest.mark.parametrize("tag", list(DETR_CASES))
def test_detr_rpe_attention_matches_reference_on_cpu(tag):
    errs = run_case(tag, "cpu")
    print(f"[detr cpu {tag}] worst {max(errs.values()):.2e}")
    assert max(errs.values()) < 2e-5, errs


def test_unsupported_options_are_refused():
    from cream_amd.detr_attention import RPEMultiheadAttention
    for kw in (dict(add_bias_kv=True), dict(add_zero_attn=True), dict(kdim=128)):
        with pytest.raises(NotImplementedError):
            RPEMultiheadAttention(256, 8, **kw)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", list(DETR_CASES))
def test_detr_rpe_attention_matches_reference_on_gpu(tag):
    from cream_amd import timing
    timing.reset()
    timing.enable(True)
    errs = run_case(tag, "cuda:0")
    timing.enable(False)
    if DETR_CASES[tag]['kw']['mode'] == 'ctx':
        assert "rpe_index_fwd" in set(timing.summary()), set(timing.summary())     # the HIP operator, not an eager gather
    print(f"[detr gpu {tag}] worst {max(errs.values()):.2e}")
    assert max(errs.values()) < 1e-3, errs

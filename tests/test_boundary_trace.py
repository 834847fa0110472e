"""The AutoFormer boundary where no reference checkout exists: the recorded sequence of
calls that the reference's UNCHANGED model/supernet_transformer.py makes into `model.module.*`
(tests/golden/autoformer_call_trace.json, written by tests/golden/make_golden.py from the reference)
must be reproduced call for call by this repository's caller, and the step it drives must reproduce
the reference-made golden step — on the CPU here and on the MI355X under `-m gpu`."""
import json
import os
import sys

import pytest
import torch

from conftest import ROOT
from helpers import check_against_fixture, config_of, load_npz

sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from fixture_utils import fill_params, make_batch, model_kwargs  # noqa: E402


def _run(device):
    from cream_amd.autoformer import supernet
    from cream_amd.dropin import trace as T
    ref_trace = json.load(open(os.path.join(ROOT, "tests", "golden", "autoformer_call_trace.json")))
    fix = load_npz("autoformer_T_step.npz")
    cfg = config_of(fix)
    rec = T.Recorder()
    with rec.patch(supernet):
        m = supernet.Vision_TransformerSuper(**model_kwargs("T"))
        fill_params(m, seed=3)
        m = m.to(device)
        m.set_sample_config(cfg)
        m.train()
        images, target = make_batch(2, seed=5)
        logits = m(images.to(device))
    got = json.loads(json.dumps(rec.events))
    assert len(got) == len(ref_trace) == 209
    for i, (a, b) in enumerate(zip(got, ref_trace)):
        assert a == b, f"boundary call {i} differs:\n ours      {a}\n reference {b}"
    loss = torch.sum(-target.to(device) * torch.log_softmax(logits, -1), -1).mean()
    loss.backward()
    grads = {k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in m.named_parameters()}
    return check_against_fixture(fix, logits, loss, grads, tol=1e-3)


def test_caller_reproduces_reference_call_trace_cpu():
    _run("cpu")


@pytest.mark.gpu
def test_caller_reproduces_reference_call_trace_and_golden_step_on_gpu():
    worst = _run("cuda:0")
    print(f"[boundary trace, MI355X] worst rel err vs the reference-made golden step: {worst:.2e}")


@pytest.mark.gpu
def test_enable_fast_path_routes_a_foreign_caller_through_the_native_stack():
    """`dropin.enable_fast_path` (what `install_autoformer` applies to the reference's
    supernet_transformer.py at import) on a stand-in caller module that, like the reference's, has a
    plain block-by-block `forward_features` and none of this repository's hooks: after the patch a
    bf16-autocast step on the device goes through `block.StackFunction` (counted) and agrees with the
    unpatched caller within the bf16 tolerance; in fp32 the caller's own code still runs."""
    import types
    import torch.nn.functional as F
    from cream_amd import dropin
    from cream_amd.autoformer import block as K, supernet as ours

    class Layer(ours.TransformerEncoderLayer):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            del self.fused, self._dp                       # the reference's layer has neither

    class Model(ours.Vision_TransformerSuper):
        def forward_features(self, x):                      # supernet_transformer.py:147-167, block by block
            B = x.shape[0]
            E = self.sample_embed_dim[0]
            x = self.patch_embed_super(x)
            x = torch.cat((self.cls_token[..., :E].expand(B, -1, -1), x), dim=1) + self.pos_embed[..., :E]
            x = F.dropout(x, p=self.sample_dropout, training=self.training)
            for blk in self.blocks:
                x = blk(x)
            x = self.norm(x)
            return torch.mean(x[:, 1:], dim=1)

    standin = types.ModuleType("standin_caller")
    standin.TransformerEncoderLayer, standin.Vision_TransformerSuper = Layer, Model
    saved = ours.TransformerEncoderLayer
    ours.TransformerEncoderLayer = Layer                    # Model.__init__ builds its blocks from this name
    try:
        m = Model(**model_kwargs("S"))
    finally:
        ours.TransformerEncoderLayer = saved
    for blk in m.blocks:
        blk.fused = False                                   # unpatched: module path
    fill_params(m, seed=11)
    m = m.to("cuda:0")
    cfg = dict(layer_num=3, embed_dim=[384] * 3, num_heads=[6, 5, 7], mlp_ratio=[3.5, 3.0, 4.0])
    m.set_sample_config(cfg)
    m.train()
    images, target = make_batch(4, seed=9)
    images, target = images.cuda(), target.cuda()

    def step():
        m.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = torch.sum(-target * torch.log_softmax(m(images).float(), -1), -1).mean()
        loss.backward()
        return float(loss), m.blocks[1].fc1.weight.grad.clone()

    l0, g0 = step()
    for blk in m.blocks:
        del blk.fused
    dropin.enable_fast_path(standin)
    calls = []
    orig = K.StackFunction.apply
    K.StackFunction.apply = staticmethod(lambda *a: (calls.append(1), orig(*a))[1])
    try:
        l1, g1 = step()
    finally:
        K.StackFunction.apply = orig
    assert calls, "the patched caller did not reach the native stack"
    assert abs(l1 - l0) / abs(l0) < 5e-3
    assert float((g1 - g0).abs().max() / g0.abs().max()) < 3e-2

"""Device-side Mixup / CutMix + soft targets (csrc/mixup.hip, one launch) against the host restatement of timm's batch-mode
Mixup (cream_amd/autoformer/data.py; timm is third-party and not vendored in the reference: AutoFormer/supernet_train.py:245-251
constructs it, supernet_engine.py:52-53 applies it — parity unpinned by the reference, the two formulations pinned to each other
here).  Same numpy seed -> same decisions (apply?, cutmix?, lambda, box) on both paths."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5, 6, 7])
@pytest.mark.parametrize("shape", [(8, 3, 32, 32), (16, 3, 224, 224)])
def test_native_mixup_matches_the_host_restatement(seed, shape):
    from cream_amd.autoformer.data import Mixup
    g = torch.Generator().manual_seed(100 + seed)
    x = torch.randn(shape, generator=g)
    t = torch.randint(0, 1000, (shape[0],), generator=g)
    mix = Mixup()                                                # the recipe: mixup 0.8, cutmix 1.0, prob 1.0, switch 0.5, smoothing 0.1
    np.random.seed(seed)
    xh, yh = mix(x.clone(), t)                                   # host tensors: the composed formulation
    np.random.seed(seed)
    xd_in = x.to(DEV)
    xd, yd = mix(xd_in, t.to(DEV))                               # device tensors: one launch, in place
    assert xd.data_ptr() == xd_in.data_ptr()
    torch.testing.assert_close(yd.cpu(), yh, rtol=0, atol=1e-7)
    # cutmix is a copy (exact); mixup is two products and a sum in the same order (exact up to the fused multiply-add contraction
    # the host's ATen kernels may or may not use)
    torch.testing.assert_close(xd.cpu(), xh, rtol=0, atol=2e-6)
    assert abs(float(yd.sum(dim=1).mean()) - 1.0) < 1e-5         # rows are distributions


def test_native_mixup_off_and_cutmix_only():
    from cream_amd.autoformer.data import Mixup
    x = torch.randn(4, 3, 16, 16)
    t = torch.tensor([1, 2, 3, 4])
    off = Mixup()
    off.mixup_enabled = False
    xd, yd = off(x.to(DEV), t.to(DEV))
    assert torch.equal(xd.cpu(), x)                              # lambda = 1: images untouched, targets = smoothed one-hot
    assert abs(float(yd[0, 1]) - (0.9 + 0.1 / 1000)) < 1e-6 and abs(float(yd[0, 2]) - 0.1 / 1000) < 1e-9
    cm = Mixup(mixup_alpha=0.0, cutmix_alpha=1.0)
    np.random.seed(11)
    xh, yh = cm(x.clone(), t)
    np.random.seed(11)
    xd, yd = cm(x.to(DEV), t.to(DEV))
    assert torch.equal(xd.cpu(), xh)                             # a pure exchange of the box between the members of a pair
    torch.testing.assert_close(yd.cpu(), yh, rtol=0, atol=1e-7)


def test_mixup_feeds_the_native_soft_ce():
    """The soft targets of the kernel are what cream_soft_ce consumes (the step: mixup_fn -> model -> criterion)."""
    from cream_amd.autoformer import engine
    from cream_amd.autoformer.data import Mixup
    np.random.seed(3)
    x = torch.randn(8, 3, 32, 32, device=DEV)
    t = torch.randint(0, 1000, (8,), device=DEV)
    _, y = Mixup()(x, t)
    logits = torch.randn(8, 1000, device=DEV, requires_grad=True)
    loss = engine.soft_target_cross_entropy(logits, y)
    want = torch.sum(-y * torch.log_softmax(logits.detach().float(), dim=-1), dim=-1).mean()
    assert abs(float(loss) - float(want)) < 1e-5 * abs(float(want))

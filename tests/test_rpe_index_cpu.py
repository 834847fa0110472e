"""CPU tests of the rpe_index slice: the oracle against the reference's own known-answer
test and against the compiled reference (oracle/_ref), and the host entry points of the
C ABI against the oracle."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT
from oracle import rpe_index_oracle as O

sys.path.insert(0, os.path.join(ROOT, "oracle"))


def _flat_gather(x, index):
    # the ground truth of the reference self-test: rpe_ops/rpe_index.py:77-78
    B, H, Lq, nb = x.shape
    Lk = index.shape[1]
    offset = torch.arange(0, Lq * nb, nb).view(-1, 1)
    return x.flatten(2)[:, :, (index.long() + offset).flatten()].view(B, H, Lq, Lk)


def test_oracle_matches_reference_selftest_shapes():
    """rpe_ops/rpe_index.py:59-100: x (128,32,50,50), random int32 index, fwd exact,
    bwd under a random mask to 5 decimals."""
    torch.manual_seed(0)
    B, H, L, nb = 128, 32, 50, 50
    x = torch.randn(B, H, L, nb)
    index = torch.randint(0, nb, (L, L)).to(torch.int)
    gt = _flat_gather(x, index)
    y = O.fwd(x.numpy(), index.numpy())
    np.testing.assert_array_equal(y, gt.numpy())
    mask = torch.randn(gt.shape)
    x2 = x.clone().requires_grad_()
    (_flat_gather(x2, index) * mask).sum().backward()
    gin = O.bwd(mask.numpy(), index.numpy(), nb)
    np.testing.assert_almost_equal(gin, x2.grad.numpy(), decimal=5)


def test_oracle_matches_compiled_reference():
    from build_ref import load_ref
    ref = load_ref()
    if ref is None:
        pytest.skip("oracle/_ref not built (needs /root/reference once)")
    torch.manual_seed(1)
    for (B, H, Lq, Lk, nb, dt) in [(2, 3, 197, 197, 50, torch.float32), (1, 2, 50, 37, 9, torch.float64),
                                   (3, 1, 5, 577, 50, torch.float16)]:
        x = torch.randn(B, H, Lq, nb).to(dt)
        index = torch.randint(0, nb, (Lq, Lk), dtype=torch.int32)
        y_ref = ref.forward_cpu(x, index)
        y = O.fwd(x.view(torch.int16).numpy() if dt == torch.float16 else x.numpy(), index.numpy())
        want = y_ref.view(torch.int16).numpy() if dt == torch.float16 else y_ref.numpy()
        np.testing.assert_array_equal(y.view(want.dtype), want)
        if dt != torch.float16:
            torch.set_num_threads(1)   # the reference's order is only defined single-threaded
            g = torch.randn(B, H, Lq, Lk).to(dt)
            gin_ref = torch.zeros(B, H, Lq, nb, dtype=dt)
            ref.backward_cpu(gin_ref, g, index)
            gin = O.bwd(g.numpy(), index.numpy(), nb)
            np.testing.assert_array_equal(gin, gin_ref.numpy())
            torch.set_num_threads(torch.get_num_threads())


@pytest.mark.parametrize("dt", [torch.float32, torch.float64, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 3, 197, 197, 50), (1, 1, 1, 1, 1), (3, 2, 7, 130, 70), (2, 2, 33, 5, 200)])
def test_host_entry_points_match_oracle(dt, shape):
    from cream_amd import rpe_index as R
    B, H, Lq, Lk, nb = shape
    torch.manual_seed(2)
    x = torch.randn(B, H, Lq, nb).to(dt)
    index = torch.randint(0, nb, (Lq, Lk), dtype=torch.int32)
    y = R.forward_cpu(x, index)
    raw = {2: torch.int16, 4: torch.int32, 8: torch.int64}[x.element_size()]
    y_or = O.fwd(x.view(raw).numpy(), index.numpy())
    np.testing.assert_array_equal(y.view(raw).numpy(), y_or)          # bit-exact
    g = torch.randn(B, H, Lq, Lk).to(dt)
    seed = torch.randn(B, H, Lq, nb).to(dt)
    gin = seed.clone()
    R.backward_cpu(gin, g, index)
    if dt in (torch.float32, torch.float64):
        want = O.bwd(g.numpy(), index.numpy(), nb, gin=seed.numpy())
        np.testing.assert_array_equal(gin.numpy(), want)               # same ascending-j order
    else:
        want = O.bwd(g.float().numpy(), index.numpy(), nb, gin=seed.float().numpy())
        np.testing.assert_array_equal(gin.float().numpy(), torch.from_numpy(want).to(dt).float().numpy())


def test_autograd_function_cpu_matches_index_autograd():
    from cream_amd.rpe_index import RPEIndexFunction
    torch.manual_seed(3)
    x1 = torch.randn(4, 3, 20, 11, requires_grad=True)
    x2 = x1.detach().clone().requires_grad_()
    index = torch.randint(0, 11, (20, 20), dtype=torch.int32)
    y = RPEIndexFunction.apply(x1, index)
    gt = _flat_gather(x2, index)
    assert torch.equal(y, gt)
    mask = torch.randn_like(gt)
    (y * mask).sum().backward()
    (gt * mask).sum().backward()
    np.testing.assert_almost_equal(x1.grad.numpy(), x2.grad.numpy(), decimal=5)


def test_error_behaviour_follows_reference():
    """AT_ASSERTM messages of rpe_index.cpp:16-20 / rpe_index_cuda.cu:62-67."""
    from cream_amd import rpe_index as R
    x = torch.randn(1, 1, 2, 3)
    with pytest.raises(RuntimeError, match="index must be Int type"):
        R.forward_cpu(x, torch.zeros(2, 2, dtype=torch.long))
    with pytest.raises(RuntimeError, match="input must be a 4D tensor"):
        R.forward_cpu(x[0], torch.zeros(2, 2, dtype=torch.int32))
    with pytest.raises(RuntimeError, match="index must be a 2D tensor"):
        R.forward_cpu(x, torch.zeros(2, dtype=torch.int32))
    with pytest.raises(RuntimeError, match="input must be a GPU tensor"):
        R.forward_gpu(x, torch.zeros(2, 2, dtype=torch.int32))
    with pytest.raises(RuntimeError, match="not implemented for"):
        R.forward_cpu(torch.zeros(1, 1, 2, 3, dtype=torch.int32), torch.zeros(2, 2, dtype=torch.int32))


def test_empty_inputs():
    from cream_amd import rpe_index as R
    y = R.forward_cpu(torch.zeros(0, 3, 4, 5), torch.zeros(4, 6, dtype=torch.int32))
    assert y.shape == (0, 3, 4, 6)
    y = R.forward_cpu(torch.zeros(2, 3, 4, 5), torch.zeros(4, 0, dtype=torch.int32))
    assert y.shape == (2, 3, 4, 0)


@pytest.mark.reference
def test_reference_irpe_runs_unchanged_on_dropin_cpu():
    """BASELINE config 1 plumbing: the reference's irpe.py imports OUR rpe_ops and its
    contextual-product iRPE gives the same numbers as its pure-PyTorch fallback."""
    import importlib
    import types
    import cream_amd.dropin as dropin
    dropin.install()
    sys.modules.setdefault("easydict", types.SimpleNamespace(EasyDict=type("EasyDict", (dict,), {
        "__getattr__": dict.__getitem__, "__setattr__": dict.__setitem__})))
    sys.path.insert(0, "/root/reference/iRPE/DeiT-with-iRPE")
    try:
        for m in ("irpe", "rpe_ops", "rpe_ops.rpe_index", "rpe_index_cpp"):
            sys.modules.pop(m, None)
        irpe = importlib.import_module("irpe")
        assert irpe.RPEIndexFunction is not None
        assert irpe.RPEIndexFunction.__module__ == "cream_amd.rpe_index"
        cfg = irpe.get_rpe_config(ratio=1.9, method="product", mode="ctx", shared_head=True, skip=1, rpe_on="k")
        _, rpe_k, _ = irpe.build_rpe(cfg, head_dim=64, num_heads=3)
        torch.manual_seed(0)
        torch.nn.init.normal_(rpe_k.lookup_table_weight, std=0.02)
        x = torch.randn(2, 3, 197, 64, requires_grad=True)
        out = rpe_k(x)
        assert out.shape == (2, 3, 197, 197)
        bucket = rpe_k._rp_bucket_buf[1]
        assert bucket.dtype == torch.int32 and int(bucket.sum()) == 941241   # SURVEY §4 checksum
        # fallback formula of irpe.py:646
        lookup = torch.matmul(x.transpose(0, 1).reshape(-1, 2 * 197, 64), rpe_k.lookup_table_weight) \
            .view(-1, 2, 197, 50).transpose(0, 1)
        off = torch.arange(0, 197 * 50, 50).view(-1, 1)
        want = lookup.flatten(2)[:, :, (bucket.long() + off).flatten()].view(2, -1, 197, 197)
        assert torch.equal(out, want)
        g = torch.randn_like(out)
        gx, gw = torch.autograd.grad(out, [x, rpe_k.lookup_table_weight], g, retain_graph=True)
        gx2, gw2 = torch.autograd.grad(want, [x, rpe_k.lookup_table_weight], g)
        torch.testing.assert_close(gx, gx2, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(gw, gw2, rtol=1e-4, atol=1e-5)
    finally:
        sys.path.remove("/root/reference/iRPE/DeiT-with-iRPE")
        for m in ("irpe", "rpe_ops", "rpe_ops.rpe_index", "rpe_index_cpp"):
            sys.modules.pop(m, None)

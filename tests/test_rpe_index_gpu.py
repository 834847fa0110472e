"""GPU parity of the rpe_index HIP kernels, called through the C ABI, against the plain-C
oracle (bit-exact: fwd for every dtype, bwd for f32/f64 thanks to the fixed ascending-j
order) plus size-independent properties at BASELINE config-4 size."""
import numpy as np
import pytest
import torch

from oracle import rpe_index_oracle as O

pytestmark = pytest.mark.gpu

RAW = {2: torch.int16, 4: torch.int32, 8: torch.int64}


def _dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


SHAPES = [
    # B, H, Lq, Lk, nb
    (2, 3, 197, 197, 50),     # DeiT 224^2 + cls, product-50
    (1, 2, 577, 577, 50),     # DeiT-B 384^2 (config 4 geometry)
    (1, 1, 1, 1, 1),          # minimum
    (3, 5, 7, 130, 70),       # nb > 64 (two lookup register slots)
    (2, 2, 33, 5, 128),       # Lk < one vector chunk, nb = 128
    (70, 1, 4, 64, 3),        # > 64 planes (two scatter waves), Lk multiple of vector width
    (1, 130, 3, 1030, 9),     # long rows -> NCHUNK > 4 fallback for f32, planes not multiple of 64
    (2, 2, 6, 9, 300),        # nb > 128 -> generic gather; large bin array
    (9, 1, 2, 25001, 6),      # rows longer than the multi-plane gather's per-thread vector budget -> one plane per workgroup;
                              # 9 planes of one alignment class and an odd row length
    (3, 7, 40, 196, 33),      # even Lk: every plane 16-byte aligned (period 1), groups of consecutive planes, last group partial
]


@pytest.mark.parametrize("dt", [torch.float32, torch.float16, torch.bfloat16, torch.float64])
@pytest.mark.parametrize("shape", SHAPES)
def test_fwd_bit_exact(shape, dt):
    from cream_amd import rpe_index as R
    B, H, Lq, Lk, nb = shape
    dev = _dev()
    g = torch.Generator().manual_seed(hash(shape) % 2**31)
    x = torch.randn(B, H, Lq, nb, generator=g).to(dt)
    index = torch.randint(0, nb, (Lq, Lk), generator=g, dtype=torch.int32)
    y = R.forward_gpu(x.to(dev), index.to(dev))
    raw = RAW[x.element_size()]
    want = O.fwd(x.view(raw).numpy(), index.numpy())
    np.testing.assert_array_equal(y.cpu().view(raw).numpy(), want)


def test_fwd_general_strides_through_the_multi_plane_gather():
    """Lookup rows with a non-unit innermost stride and padded rows (the staged-table path with the per-element index
    arithmetic), planes spread over several alignment classes."""
    from cream_amd import rpe_index as R
    dev = _dev()
    B, H, L, nb = 3, 5, 57, 11
    torch.manual_seed(1)
    base = torch.randn(B, H, L + 2, 2 * nb + 3)
    view = base.to(dev)[:, :, 1:L + 1, 1:2 * nb + 1:2]
    assert view.shape == (B, H, L, nb) and view.stride(3) == 2
    index = torch.randint(0, nb, (L, L), dtype=torch.int32)
    y = R.forward_gpu(view, index.to(dev))
    want = base[:, :, 1:L + 1, 1:2 * nb + 1:2].contiguous()[:, :, torch.arange(L)[:, None], index.long()]
    assert torch.equal(y.cpu(), want)


def test_fwd_transposed_view_input():
    """iRPE hands over lookup_table as a transposed view (irpe.py:639-642):
    strides (L*nb, B*L*nb, nb, 1)."""
    from cream_amd import rpe_index as R
    dev = _dev()
    B, H, L, nb = 5, 3, 197, 50
    torch.manual_seed(0)
    base = torch.randn(H, B, L, nb)
    index = torch.randint(0, nb, (L, L), dtype=torch.int32)
    view = base.to(dev).transpose(0, 1)
    assert not view.is_contiguous() and view.stride() == (L * nb, B * L * nb, nb, 1)
    y = R.forward_gpu(view, index.to(dev))
    want = O.fwd_strided_f32(base.numpy().ravel(), (B, H, L, nb), view.stride(), index.numpy())
    np.testing.assert_array_equal(y.cpu().numpy(), want)
    # generic element stride on the bucket axis as well
    wide = torch.randn(B, H, L, 2 * nb)
    v2 = wide.to(dev)[..., ::2]
    y2 = R.forward_gpu(v2, index.to(dev))
    want2 = O.fwd(wide[..., ::2].contiguous().numpy(), index.numpy())
    np.testing.assert_array_equal(y2.cpu().numpy(), want2)


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
@pytest.mark.parametrize("shape", SHAPES)
def test_bwd_bit_exact_fixed_order(shape, dt):
    from cream_amd import rpe_index as R
    B, H, Lq, Lk, nb = shape
    dev = _dev()
    g = torch.Generator().manual_seed(hash(shape) % 2**31 + 1)
    gout = torch.randn(B, H, Lq, Lk, generator=g).to(dt)
    seed = torch.randn(B, H, Lq, nb, generator=g).to(dt)     # accumulate INTO grad_input
    index = torch.randint(0, nb, (Lq, Lk), generator=g, dtype=torch.int32)
    gin = seed.to(dev)
    R.backward_gpu(gin, gout.to(dev), index.to(dev))
    want = O.bwd(gout.numpy(), index.numpy(), nb, gin=seed.numpy())
    np.testing.assert_array_equal(gin.cpu().numpy(), want)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape", SHAPES[:6])
def test_bwd_16bit_accumulates_in_f32(shape, dt):
    from cream_amd import rpe_index as R
    B, H, Lq, Lk, nb = shape
    dev = _dev()
    g = torch.Generator().manual_seed(7)
    gout = torch.randn(B, H, Lq, Lk, generator=g).to(dt)
    index = torch.randint(0, nb, (Lq, Lk), generator=g, dtype=torch.int32)
    gin = torch.zeros(B, H, Lq, nb, dtype=dt, device=dev)
    R.backward_gpu(gin, gout.to(dev), index.to(dev))
    want = torch.from_numpy(O.bwd(gout.float().numpy(), index.numpy(), nb)).to(dt)
    np.testing.assert_array_equal(gin.cpu().float().numpy(), want.float().numpy())


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_bwd_overwrite_mode_ignores_garbage(dt):
    """accumulate=0 (C-ABI extension): grad_input need not be initialised."""
    from cream_amd import rpe_index as R
    dev = _dev()
    g = torch.Generator().manual_seed(11)
    gout = torch.randn(3, 5, 197, 197, generator=g).to(dt)
    index = torch.randint(0, 50, (197, 197), generator=g, dtype=torch.int32)
    gin = torch.full((3, 5, 197, 50), float("nan"), dtype=dt, device=dev)
    R.backward_gpu(gin, gout.to(dev), index.to(dev), accumulate=False)
    want = torch.from_numpy(O.bwd(gout.float().numpy(), index.numpy(), 50)).to(dt)
    np.testing.assert_array_equal(gin.cpu().float().numpy(), want.float().numpy())


def test_bwd_is_reproducible():
    from cream_amd import rpe_index as R
    dev = _dev()
    torch.manual_seed(3)
    gout = torch.randn(8, 12, 197, 197, device=dev)
    index = torch.randint(0, 50, (197, 197), dtype=torch.int32, device=dev)
    outs = []
    for _ in range(3):
        gin = torch.zeros(8, 12, 197, 50, device=dev)
        R.backward_gpu(gin, gout, index)
        outs.append(gin)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


def test_autograd_function_matches_reference_selftest():
    """rpe_ops/rpe_index.py:59-100 on the GPU: fwd exact, bwd 5 decimals vs index autograd."""
    from cream_amd.rpe_index import RPEIndexFunction
    dev = _dev()
    torch.manual_seed(0)
    B, H, L, nb = 128, 32, 50, 50
    x = torch.randn(B, H, L, nb, device=dev)
    index = torch.randint(0, nb, (L, L), device=dev).to(torch.int)
    offset = torch.arange(0, L * nb, nb, device=dev).view(-1, 1)
    x1 = x.clone().requires_grad_()
    x2 = x.clone().requires_grad_()
    y = RPEIndexFunction.apply(x1, index)
    gt = x2.flatten(2)[:, :, (index + offset).flatten()].view(B, H, L, L)
    assert torch.equal(y, gt)
    mask = torch.randn_like(gt)
    (gt * mask).sum().backward()
    (y * mask).sum().backward()
    np.testing.assert_almost_equal(x1.grad.cpu().numpy(), x2.grad.cpu().numpy(), decimal=5)


def test_full_size_properties_config4():
    """BASELINE config 4 (B=64,H=12,L=577,nb=50; 1.02 GB output): too big for the scalar
    oracle, so check size-independent properties on the device:
      * fwd equals torch's own gather (exact), on the transposed-view input;
      * <fwd(x), g> == <x, bwd(g)>   (adjointness, fp64 accumulation of both sides);
      * a one-hot probe: bwd(fwd-mask) counts bucket populations exactly."""
    from cream_amd import rpe_index as R
    dev = _dev()
    B, H, L, nb = 64, 12, 577, 50
    torch.manual_seed(4)
    idx_cpu = torch.randint(0, nb, (L, L), dtype=torch.int32)
    index = idx_cpu.to(dev)
    x = torch.randn(H, B, L, nb, device=dev).transpose(0, 1)
    y = R.forward_gpu(x, index)
    want = torch.gather(x, 3, index.long().unsqueeze(0).unsqueeze(0).expand(B, H, L, L))
    assert torch.equal(y, want)
    del want
    g = torch.randn(B, H, L, L, device=dev)
    gin = torch.zeros(B, H, L, nb, device=dev)
    R.backward_gpu(gin, g, index)
    lhs = (y.double() * g.double()).sum().item()
    rhs = (x.double() * gin.double()).sum().item()
    assert abs(lhs - rhs) <= 1e-6 * max(1.0, abs(lhs))   # gin is f32-rounded
    del y, g
    ones = torch.ones(2, 1, L, L, device=dev)
    cnt = torch.zeros(2, 1, L, nb, device=dev)
    R.backward_gpu(cnt, ones, index)
    pop = torch.stack([(idx_cpu == u).sum(1) for u in range(nb)], 1).float()
    assert torch.equal(cnt[0, 0].cpu(), pop) and torch.equal(cnt[1, 0].cpu(), pop)


def test_runs_on_current_stream_without_sync():
    from cream_amd import rpe_index as R
    dev = _dev()
    s = torch.cuda.Stream()
    x = torch.randn(4, 3, 197, 50, device=dev)
    index = torch.randint(0, 50, (197, 197), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        y = R.forward_gpu(x, index)
    s.synchronize()
    assert torch.equal(y, torch.gather(x, 3, index.long().expand(4, 3, 197, 197)))


def test_gpu_error_behaviour():
    from cream_amd import rpe_index as R
    dev = _dev()
    x = torch.randn(1, 1, 4, 3, device=dev)
    idx = torch.zeros(4, 6, dtype=torch.int32, device=dev)
    with pytest.raises(RuntimeError, match="index should be contiguous"):
        R.forward_gpu(x, idx[:, ::2])
    with pytest.raises(RuntimeError, match="index must be a GPU tensor"):
        R.forward_gpu(x, idx.cpu())
    with pytest.raises(RuntimeError, match="index must be Int type"):
        R.forward_gpu(x, idx.long())

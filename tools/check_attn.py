"""GPU check of the fused attention kernels against the bucketed HIP path and an fp64
dense restatement (development aid; the pytest versions live in tests/test_attn_gpu.py)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cream_amd.autoformer import attention_op, fused_attention
from cream_amd.autoformer.modules import relative_index_tables


def dense_ref(qkv, tkv, tkh, tvv, tvh, iv, ih, scale):
    q, k, v = qkv.double().permute(2, 0, 3, 1, 4).unbind(0)
    rk = (tkv.double()[iv.long()] + tkh.double()[ih.long()])          # (N, N, D)
    rv = (tvv.double()[iv.long()] + tvh.double()[ih.long()])
    a = (q @ k.transpose(-1, -2) + torch.einsum('bhid,ijd->bhij', q, rk)) * scale
    p = a.softmax(-1)
    o = p @ v + torch.einsum('bhij,ijd->bhid', p, rv)
    return o.transpose(1, 2)


def run(B, H, side, dtype, mr=14, bwd=True, seed=0):
    dev = torch.device('cuda')
    N = side * side + 1
    g = torch.Generator(device='cpu').manual_seed(seed)
    qkv = torch.randn(B, N, 3, H, 64, generator=g).to(dev)
    tabs = [(torch.randn(2 * mr + 2, 64, generator=g) * 0.5).to(dev) for _ in range(4)]
    go = torch.randn(B, N, H, 64, generator=g).to(dev)
    iv, ih = relative_index_tables(N, mr, dev)
    scale = 0.125
    leaves = [qkv.clone().requires_grad_()] + [t.clone().requires_grad_() for t in tabs]
    ref = dense_ref(leaves[0], *leaves[1:], iv, ih, scale)
    if bwd:
        ref.backward(go.double())
    refg = [x.grad for x in leaves]
    x = [qkv.to(dtype).requires_grad_()] + [t.clone().requires_grad_() for t in tabs]
    out = fused_attention.attention_rpe2d_fused(x[0], *x[1:], scale, mr)
    torch.cuda.synchronize()
    def rel(a, b):
        return float((a.double() - b.double()).abs().max() / b.double().abs().max())
    print(f"B{B} H{H} N{N} {dtype}: out rel {rel(out, ref):.3e}", end='')
    if bwd:
        out.backward(go.to(dtype))
        torch.cuda.synchronize()
        names = ['dqkv', 'dtkv', 'dtkh', 'dtvv', 'dtvh']
        for n, a, b in zip(names, [t.grad for t in x], refg):
            print(f"  {n} {rel(a, b):.3e}", end='')
    print()


if __name__ == '__main__':
    bwd = '--fwd' not in sys.argv
    for dtype in (torch.float32, torch.bfloat16):
        run(2, 3, 14, dtype, bwd=bwd)
        run(1, 2, 7, dtype, bwd=bwd)
        run(1, 1, 15, dtype, mr=14, bwd=bwd)
        run(1, 2, 5, dtype, mr=3, bwd=bwd)       # clamping active

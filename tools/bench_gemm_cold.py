"""Library GEMM on the supernet's shapes with operands rotated through > 256 MB (Infinity Cache cold)."""
import torch, time
dev = torch.device('cuda')
M = 25216
def t(fn, n):
    for i in range(2): fn(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): fn(i)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
R = 12
for (E, F) in [(384, 1344), (448, 1792), (320, 960)]:
    xs = [torch.randn(M, E, device=dev, dtype=torch.bfloat16) for _ in range(R)]
    ys = [torch.empty(M, F, device=dev, dtype=torch.bfloat16) for _ in range(R)]
    dys = [torch.randn(M, F, device=dev, dtype=torch.bfloat16) for _ in range(R)]
    dxs = [torch.empty(M, E, device=dev, dtype=torch.bfloat16) for _ in range(R)]
    W = torch.randn(1792, 448, device=dev, dtype=torch.bfloat16)[:F, :E]
    fl = 2 * M * E * F
    s = t(lambda i: torch.mm(xs[i % R], W.t(), out=ys[i % R]), 36); print(f"E{E} F{F} fwd NT cold  {s*1e6:7.1f} us {fl/s/1e12:6.1f} TF/s")
    s = t(lambda i: torch.mm(dys[i % R], W, out=dxs[i % R]), 36); print(f"E{E} F{F} dgrad NN cold {s*1e6:7.1f} us {fl/s/1e12:6.1f} TF/s")
    s = t(lambda i: torch.mm(dys[i % R].t(), xs[i % R]), 36); print(f"E{E} F{F} wgrad TN cold {s*1e6:7.1f} us {fl/s/1e12:6.1f} TF/s")
    s = t(lambda i: torch.bmm(dys[i % R].view(8, M // 8, F).transpose(1, 2), xs[i % R].view(8, M // 8, E)).sum(0), 36); print(f"E{E} F{F} wgrad splitK8 cold {s*1e6:7.1f} us {fl/s/1e12:6.1f} TF/s")
    s = t(lambda i: ys[i % R].copy_(dys[(i + 1) % R]), 36); print(f"   copy {M}x{F} bf16 cold {s*1e6:7.1f} us {2*M*F*2/s/1e9:6.0f} GB/s")

"""How fast is the library GEMM (hipBLASLt via torch) on the supernet's shapes?  (development aid)"""
import torch, time
dev = torch.device('cuda')
M = 25216

def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n

def report(name, flops, sec):
    print(f"{name:60s} {sec*1e6:8.1f} us  {flops/sec/1e12:7.1f} TF/s")

for (E, F) in [(384, 1344), (448, 1792), (320, 960)]:
    x = torch.randn(M, E, device=dev, dtype=torch.bfloat16)
    Wsup = torch.randn(1792, 448, device=dev, dtype=torch.bfloat16)
    Wc = Wsup[:F, :E].contiguous()
    Wv = Wsup[:F, :E]
    b = torch.randn(F, device=dev, dtype=torch.bfloat16)
    fl = 2 * M * E * F
    report(f"fwd NT contiguous W  E{E} F{F}", fl, t(lambda: x @ Wc.t()))
    report(f"fwd NT strided W (ld 448)", fl, t(lambda: x @ Wv.t()))
    report(f"fwd F.linear strided + bias", fl, t(lambda: torch.nn.functional.linear(x, Wv, b)))
    dy = torch.randn(M, F, device=dev, dtype=torch.bfloat16)
    report(f"dgrad NN  dy @ W contiguous", fl, t(lambda: dy @ Wc))
    report(f"dgrad NN  dy @ W strided", fl, t(lambda: dy @ Wv))
    report(f"wgrad TN  dy^T @ x", fl, t(lambda: dy.t() @ x))
    S = 8
    dy3 = dy.view(S, M // S, F); x3 = x.view(S, M // S, E)
    report(f"wgrad TN  split-K bmm x{S} + sum", fl, t(lambda: torch.bmm(dy3.transpose(1, 2), x3).sum(0)))
    xf = x.float(); 
    report(f"fp32 residual add (M x E)", 0 + 1, t(lambda: xf + xf))
    print()

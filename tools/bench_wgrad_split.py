"""Weight-gradient GEMM dW = dY^T X over M = 25216 tokens: time per split-K factor (library default
kernel selection vs the committed offline table for split 8), and a single fp32-output GEMM."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cream_amd.autoformer import engine
dev = torch.device('cuda')
M = 25216
shapes = [(1344, 384), (384, 1344), (384, 384), (1152, 384), (1792, 448), (448, 1792), (960, 320)]
use_table = '--table' in sys.argv
if use_table:
    print('offline table loaded:', engine.enable_gemm_selection('S', 128))


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for out, inn in shapes:
    dy = torch.randn(M, out, device=dev).bfloat16()
    x = torch.randn(M, inn, device=dev).bfloat16()
    row = []
    for s in (1, 2, 4, 8, 16, 32, 64):
        if M % s:
            continue
        f = lambda: torch.bmm(dy.view(s, M // s, -1).transpose(1, 2), x.view(s, M // s, -1))
        us = timeit(f)
        row.append(f"s{s}:{us:6.1f}us({2 * M * out * inn / us / 1e6:4.0f}TF)")
    print(f"dW {out}x{inn}: " + "  ".join(row), flush=True)

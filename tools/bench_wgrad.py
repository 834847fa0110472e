"""Standalone timing of the weight-gradient kernels at supernet-S block shapes (M = 25,216): the grouped stream-K launch
(cream_wgrad_group) against the four split-K launches (cream_linear_wgrad_parts) of round 2."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from cream_amd.autoformer import block as K

DEV = "cuda:0"


def med(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2] * 1e3


def main():
    M = 25216
    for E, H, F in ((320, 5, 960), (384, 6, 1344), (448, 7, 1792)):
        Q = 64 * H
        shapes = [(E, F, 0), (F, E, 0), (E, Q, 0), (3 * Q, E, Q)]
        probs = []
        for N, Kd, inter in shapes:
            dy = torch.randn(M, N, device=DEV).to(torch.bfloat16)
            x = torch.randn(M, Kd, device=DEV).to(torch.bfloat16)
            w = torch.nn.Parameter(torch.zeros(N, Kd, device=DEV))
            b = torch.nn.Parameter(torch.zeros(N, device=DEV)) if inter else None
            w.grad = torch.zeros_like(w)
            if b is not None:
                b.grad = torch.zeros_like(b)
            probs.append((dy, x, w, b, inter))
        flops = sum(2 * M * n * k for n, k, _ in shapes)
        t_group = med(lambda: K.wgrad_group(probs))
        t_parts = med(lambda: [K.linear_wgrad_parts(dy, x, want_bias=b is not None) for dy, x, w, b, _ in probs])
        print(json.dumps(dict(E=E, H=H, F=F, group_us=round(t_group, 1), group_tflops=round(flops / t_group / 1e6, 1),
                              parts4_us=round(t_parts, 1), parts4_tflops=round(flops / t_parts / 1e6, 1))))


if __name__ == "__main__":
    main()

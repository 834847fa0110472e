"""Micro-benchmark of the rpe_index HIP kernels at BASELINE config 4
(DeiT-B-384 + iRPE product-ctx: B=64, H=12, L=577, nb=50), f32 and bf16.

Prints achieved GB/s against the ALGORITHMIC bytes of SURVEY §8(d):
  fwd: B*H*Lq*nb*s (lookup rows) + Lq*Lk*4 (index, once) + B*H*Lq*Lk*s (output)
  bwd: B*H*Lq*Lk*s (grad_output) + Lq*Lk*4 + B*H*Lq*nb*s (grad_input written)
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from cream_amd import rpe_index as R

PEAK_HBM_GBS = 8000.0


def time_kernel(fn, iters, warmup):
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
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=64)
    ap.add_argument("--H", type=int, default=12)
    ap.add_argument("--L", type=int, default=577)
    ap.add_argument("--nb", type=int, default=50)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    B, H, L, nb = a.B, a.H, a.L, a.nb
    index = torch.randint(0, nb, (L, L), dtype=torch.int32, device=dev)
    out = []
    for dt in (torch.float32, torch.bfloat16):
        s = torch.empty((), dtype=dt).element_size()
        x = torch.randn(H, B, L, nb, device=dev).to(dt).transpose(0, 1)      # iRPE's transposed view
        g = torch.randn(B, H, L, L, device=dev).to(dt)
        gin = torch.zeros(B, H, L, nb, device=dev, dtype=dt)
        bytes_alg = B * H * L * nb * s + L * L * 4 + B * H * L * L * s
        med, best = time_kernel(lambda: R.forward_gpu(x, index), a.iters, a.warmup)
        out.append(dict(kernel="rpe_index_fwd", dtype=str(dt), ms=med, ms_best=best,
                        bytes=bytes_alg, GBps=bytes_alg / med / 1e6, frac=bytes_alg / med / 1e6 / PEAK_HBM_GBS))
        med, best = time_kernel(lambda: R.backward_gpu(gin, g, index, accumulate=False), a.iters, a.warmup)
        out.append(dict(kernel="rpe_index_bwd", dtype=str(dt), ms=med, ms_best=best,
                        bytes=bytes_alg, GBps=bytes_alg / med / 1e6, frac=bytes_alg / med / 1e6 / PEAK_HBM_GBS))
        # calibration on the same buffers: what this box sustains for a pure write / a copy
        y = torch.empty(B, H, L, L, device=dev, dtype=dt)
        med, _ = time_kernel(lambda: y.fill_(1.0), a.iters, a.warmup)
        out.append(dict(kernel="calib_fill", dtype=str(dt), ms=med, bytes=y.numel() * s, GBps=y.numel() * s / med / 1e6))
        med, _ = time_kernel(lambda: y.copy_(g), a.iters, a.warmup)
        out.append(dict(kernel="calib_copy", dtype=str(dt), ms=med, bytes=2 * y.numel() * s, GBps=2 * y.numel() * s / med / 1e6))
        med, _ = time_kernel(lambda: g.sum(), a.iters, a.warmup)
        out.append(dict(kernel="calib_read_sum", dtype=str(dt), ms=med, bytes=y.numel() * s, GBps=y.numel() * s / med / 1e6))
        del y
        del x, g, gin
    for o in out:
        print(json.dumps(o))


if __name__ == "__main__":
    main()

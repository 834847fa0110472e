"""Standalone timing of the projection + residual add + LayerNorm kernel (csrc/gemm_ln.hip) against the two-kernel path
(cream_linear_fwd + cream_add_ln_fwd) at the shapes of the AutoFormer-S step (M = 128 x 197).  Median of 20 launches."""
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cream_amd.autoformer import block as K  # noqa: E402

DEV = "cuda:0"


def med(fn, n=20):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    """Buffers rotate over NSETS sets (> 256 MB in total) so that no launch finds its operands in the memory-side cache —
    the condition inside the training step; `--warm` reuses one set (what a naive loop measures)."""
    M, N = 128 * 197, 197
    nsets = 1 if "--warm" in sys.argv else 6
    g = torch.Generator(device=DEV).manual_seed(0)
    for E, Kd in ((320, 320), (384, 384), (448, 448), (320, 960), (384, 1344), (448, 1792)):
        sets = []
        for _ in range(nsets):
            sets.append(dict(a=torch.randn(M, Kd, device=DEV, generator=g).bfloat16(), x=torch.randn(M, E, device=DEV, generator=g),
                             out=torch.empty(M, E, device=DEV, dtype=torch.bfloat16)))
        w = (torch.randn(E, Kd, device=DEV, generator=g) * Kd ** -0.5).bfloat16()
        b = torch.randn(E, device=DEV, generator=g).bfloat16()
        sc = (torch.rand(128, device=DEV, generator=g) > 0.1).float() / 0.9
        gamma, beta = torch.randn(E, device=DEV, generator=g), torch.randn(E, device=DEV, generator=g)
        it = [0]

        def nxt():
            it[0] += 1
            return sets[it[0] % nsets]

        def gemm():
            s_ = nxt(); K.linear_fwd(s_["a"], w, b, E, Kd, out=s_["out"])

        def ln():
            s_ = nxt(); K.add_ln_fwd(s_["x"], s_["out"], sc, N, gamma, beta, 1e-5)

        def both():
            s_ = nxt(); K.linear_fwd(s_["a"], w, b, E, Kd, out=s_["out"]); K.add_ln_fwd(s_["x"], s_["out"], sc, N, gamma, beta, 1e-5)

        def fused():
            s_ = nxt(); K.linear_add_ln_fwd(s_["a"], w, b, s_["x"], sc, N, gamma, beta, 1e-5, Kd)

        t_g, t_l, t_b, t_f = med(gemm), med(ln), med(both), med(fused)
        nbytes = M * E * 10 + M * Kd * 2
        print(json.dumps(dict(E=E, K=Kd, sets=nsets, gemm_us=round(t_g, 1), add_ln_us=round(t_l, 1), two_kernels_us=round(t_b, 1),
                              fused_us=round(t_f, 1), fused_TFLOPs=round(2 * M * E * Kd / t_f / 1e6, 1),
                              fused_GBps=round(nbytes / t_f / 1e3, 1))), flush=True)


if __name__ == "__main__":
    main()

"""BASELINE config 4: DeiT-base-384 + iRPE product / contextual (rpe_ops fwd/bwd), B = 64, H = 12,
L = 577, head_dim 64, 50 buckets — ONE RPEAttention layer forward + backward on the MI355X
(after the standalone rpe_index micro-benchmark, "full RPEAttention fwd/bwd for
rpe_on in {k, qkv}").  Reports ms per fwd+bwd, the share of the rpe_index kernels (HIP events on
the launch stream, cream_amd.timing) and their achieved HBM GB/s against the algorithmic bytes.

    python tools/bench_irpe_attention.py > gpurun_out/irpe_attention.jsonl
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from cream_amd import timing
from cream_amd.irpe import get_rpe_config
from cream_amd.rpe_attention import RPEAttention

dev = torch.device("cuda")
B, L, C, H = 64, 577, 768, 12
for rpe_on in ("k", "qkv"):
    for dtype in (torch.bfloat16, torch.float32):
        torch.manual_seed(0)
        cfg = get_rpe_config(ratio=1.9, method="product", mode="ctx", shared_head=True, skip=1, rpe_on=rpe_on)
        m = RPEAttention(C, num_heads=H, qkv_bias=True, rpe_config=cfg).to(dev)
        x = torch.randn(B, L, C, device=dev, requires_grad=True)
        g = torch.randn(B, L, C, device=dev)

        def step():
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=dtype == torch.bfloat16):
                y = m(x)
            y.backward(g)
            x.grad = None
            for p in m.parameters():
                p.grad = None

        for _ in range(3):
            step()
        torch.cuda.synchronize()
        timing.reset()
        timing.enable(True, only=("rpe_index_fwd", "rpe_index_bwd", "irpe_attn_fwd", "irpe_attn_bwd"))
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 10
        a.record()
        for _ in range(n):
            step()
        b.record()
        torch.cuda.synchronize()
        timing.enable(False)
        ks = timing.summary()
        rec = dict(workload=f"RPEAttention fwd+bwd, DeiT-B-384 iRPE product-ctx rpe_on={rpe_on}", B=B, H=H, L=L,
                   dtype=str(dtype).split(".")[-1], ms_per_fwd_bwd=round(a.elapsed_time(b) / n, 3),
                   kernels={k: dict(launches=v["launches"], avg_us=round(v["avg_ms"] * 1e3, 1),
                                    total_ms_per_iter=round(v["total_ms"] / n, 3),
                                    GBps=round(v["bytes"] / (v["total_ms"] * 1e-3) / 1e9, 1) if v["bytes"] else None,
                                    TFLOPs=round(v["flops"] / (v["total_ms"] * 1e-3) / 1e12, 1) if v.get("flops") else None)
                            for k, v in sorted(ks.items())})
        print(json.dumps(rec), flush=True)

"""Forward / backward kernel time of the fused iRPE attention at config 4 by rpe subset (which term costs what)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cream_amd import timing, irpe as I, irpe_fused

dev = "cuda:0"
B, L, H = 64, 577, 12
for rpe_on in (sys.argv[1].split(",") if len(sys.argv) > 1 else ("", "k", "q", "v", "qk", "kv", "qkv")):   # "none" = no rpe
    rpe_on = "" if rpe_on == "none" else rpe_on
    torch.manual_seed(0)
    mods = [None, None, None]
    if rpe_on:
        cfg = I.get_rpe_config(ratio=1.9, method="product", mode="ctx", shared_head=True, skip=1, rpe_on=rpe_on)
        mods = [m.to(dev) if m is not None else None for m in I.build_rpe(cfg, head_dim=64, num_heads=H)]
    qkv = (0.8 * torch.randn(B, L, 3, H, 64, device=dev)).to(torch.bfloat16).requires_grad_()
    gy = torch.randn(B, L, H * 64, device=dev).to(torch.bfloat16)
    for _ in range(3):
        y = irpe_fused.attention(qkv, 0.125, *mods); y.backward(gy)
    torch.cuda.synchronize()
    timing.reset(); timing.enable(True, only=("irpe_attn_fwd", "irpe_attn_bwd"))
    for _ in range(10):
        y = irpe_fused.attention(qkv, 0.125, *mods); y.backward(gy)
    torch.cuda.synchronize(); timing.enable(False)
    ks = timing.summary()
    print(json.dumps({"rpe_on": rpe_on or "none", **{k: round(v["avg_ms"] * 1e3, 1) for k, v in sorted(ks.items())}}))

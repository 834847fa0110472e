"""BASELINE config 4 as a whole model: DeiT-base-384 + iRPE (product, contextual), batch 64, one training step
(forward + backward + AdamW) under bf16 autocast on the MI355X — the fused iRPE attention (csrc/irpe_attn.hip) against the
composed path on the HIP rpe_index operator (CREAM_IRPE_FUSED=0: what the reference's module structure does), same process.
The MLP / LayerNorm / linears of the model are the framework's (outside SURVEY 8's path); the attention core is ours.

    python tools/bench_deit_irpe.py > gpurun_out/deit_irpe.jsonl
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from cream_amd.rpe_attention import deit_irpe

dev = torch.device("cuda")
B = int(os.environ.get("DEIT_BATCH", "64"))
ONLY = os.environ.get("DEIT_ONLY")                 # e.g. "k1": rpe on k, own kernels only (profiling)
for rpe_on in ("k", "qkv"):
    for fused, native in (("1", "1"), ("1", "0"), ("0", "0")):
        if ONLY and (rpe_on, native) != (ONLY[:-1], ONLY[-1]):
            continue
        os.environ["CREAM_IRPE_FUSED"], os.environ["CREAM_DEIT_NATIVE"] = fused, native
        torch.manual_seed(0)
        model = deit_irpe("base", img_size=384, rpe_on=rpe_on, drop_path_rate=float(os.environ.get("DEIT_DROP_PATH", "0"))).to(dev)
        opt = torch.optim.AdamW(model.parameters(), lr=1e-4, fused=True)
        x = torch.randn(B, 3, 384, 384, device=dev)
        y = torch.randint(0, 1000, (B,), device=dev)

        def step():
            with torch.autocast("cuda", dtype=torch.bfloat16):
                loss = torch.nn.functional.cross_entropy(model(x).float(), y)
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            return loss

        for _ in range(3):
            step()
        torch.cuda.synchronize()
        n = 8
        t0 = time.perf_counter()
        for _ in range(n):
            loss = step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        print(json.dumps({"workload": f"DeiT-base-384 + iRPE product-ctx rpe_on={rpe_on}, train step, bf16 autocast", "batch": B, "L": 577,
                          "blocks": ("own kernels end to end (cream_amd/deit_native.py)" if native == "1" else "framework linears / LayerNorm / GELU"),
                          "attention": "fused (csrc/irpe_attn.hip)" if fused == "1" else "composed (rpe_index gather / scatter-add, (B,H,L,L) tensors)",
                          "drop_path": float(os.environ.get("DEIT_DROP_PATH", "0")), "ms_per_step": round(ms, 2), "images_per_s": round(B / ms * 1e3, 1), "loss": round(float(loss), 4),
                          "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}), flush=True)
        del model, opt
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()
os.environ.pop("CREAM_IRPE_FUSED", None)
os.environ.pop("CREAM_DEIT_NATIVE", None)

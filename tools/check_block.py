"""GPU check: fused block (bf16) vs the module path (bf16 autocast) and vs fp32 (development aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cream_amd.autoformer import engine

dev = torch.device('cuda')
torch.manual_seed(0)
model = engine.build_supernet('S', drop_path_rate=0.0, depth=3).to(dev)
cfg = dict(layer_num=3, embed_dim=[384] * 3, num_heads=[6, 5, 7], mlp_ratio=[3.5, 4.0, 3.0])
model.set_sample_config(cfg); model.train()
B = 8
x = torch.randn(B, 3, 224, 224, device=dev)
t = torch.softmax(torch.randn(B, 1000, device=dev), -1)

def run(fused, amp):
    for b in model.blocks: b.fused = fused
    model.zero_grad(set_to_none=True)
    with torch.autocast('cuda', dtype=torch.bfloat16, enabled=amp):
        loss = engine.soft_target_cross_entropy(model(x), t)
    loss.backward()
    return float(loss), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}

l32, g32 = run(False, False)
lm, gm = run(False, True)
lf, gf = run(True, True)
print(f"loss fp32 {l32:.6f} module-bf16 {lm:.6f} fused-bf16 {lf:.6f}")
def rel(a, b): return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))
worst_m = worst_f = 0
for k in g32:
    em, ef = rel(gm[k], g32[k]), rel(gf[k], g32[k])
    worst_m, worst_f = max(worst_m, em), max(worst_f, ef)
    if ef > 3 * em + 2e-2: print(f"  {k}: module {em:.3e} fused {ef:.3e}")
print(f"worst rel err vs fp32: module-bf16 {worst_m:.3e}  fused-bf16 {worst_f:.3e}")
assert set(gf) == set(g32), set(g32) ^ set(gf)

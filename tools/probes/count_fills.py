"""Which host code issues the per-step zero fills / copies?  Patches the usual suspects and counts by call site."""
import collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
from cream_amd import comm
from cream_amd.autoformer import engine

dev = torch.device("cuda")
model = engine.build_supernet("S").to(dev)
import yaml  # noqa
choices = dict(num_heads=[5, 6, 7], mlp_ratio=[3.0, 3.5, 4.0], embed_dim=[320, 384, 448], depth=[12, 13, 14])
opt = engine.build_optimizer(model, batch_size=128)
red = comm.GradReducer(model)
tr = engine.SupernetTrainer(model, opt, choices, red)
x = torch.randn(128, 3, 224, 224, device=dev)
t = torch.softmax(torch.randn(128, 1000, device=dev), -1)
tr.start_epoch(0)
for _ in range(3):
    tr.step(x, t)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    tr.step(x, t)
torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::zero_", "aten::fill_", "aten::zeros", "aten::zeros_like", "aten::copy_", "aten::new_zeros"):
        st = [s for s in (ev.stack or []) if "cream_amd" in s or "bench" in s or "autograd" in s]
        cnt[(ev.name, st[0] if st else (ev.stack[0] if ev.stack else "?"), str(ev.input_shapes)[:60])] += 1
for k, v in cnt.most_common(40):
    print(v, k)

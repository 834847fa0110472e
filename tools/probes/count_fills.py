"""Which host code issues the per-step fill / copy kernels?  One profiled step; every CPU op whose kernels include a
fill or a copy is counted by (op, shapes, first cream_amd / bench frame of its stack)."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import ProfilerActivity, profile

from cream_amd import comm
from cream_amd.autoformer import engine

dev = torch.device("cuda")
model = engine.build_supernet("S").to(dev)
opt = engine.build_optimizer(model, batch_size=128)
red = comm.GradReducer(model)
tr = engine.SupernetTrainer(model, opt, engine.SEARCH_SPACES["S"]["choices"], red)
x = torch.randn(128, 3, 224, 224, device=dev)
t = torch.softmax(torch.randn(128, 1000, device=dev), -1)
tr.start_epoch(0)
for _ in range(3):
    tr.step(x, t)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    tr.step(x, t)
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    ks = [k.name for k in (getattr(ev, "kernels", None) or [])]
    hit = [k for k in ks if "FillFunctor" in k or "copyBuffer" in k or "copy_kernel" in k]
    if not hit:
        continue
    st = [s for s in (ev.stack or []) if "cream_amd" in s or "bench" in s or "count_fills" in s]
    cnt[(ev.name, str(ev.input_shapes)[:50], st[0][-90:] if st else "?", hit[0][:40])] += 1
for k, v in cnt.most_common(40):
    print(v, k)

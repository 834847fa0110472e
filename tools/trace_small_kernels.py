"""Which host code launches the small framework kernels of the step (fills, device-to-device copies, elementwise)?
torch.profiler with Python stacks over a few steps of the bench configuration; prints, per kernel name, the launch count
per step and the innermost cream_amd / bench frames of its launch sites.
usage (GPU box): python tools/trace_small_kernels.py [steps]"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cream_amd import comm  # noqa: E402
from cream_amd.autoformer import engine  # noqa: E402


def main(steps=4):
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = engine.build_supernet("S", drop_path_rate=0.1).to(dev)
    opt = engine.build_optimizer(model, lr=5e-4, batch_size=128, world_size=1)
    trainer = engine.SupernetTrainer(model, opt, engine.SEARCH_SPACES["S"]["choices"], comm.GradReducer(model), amp_dtype=torch.bfloat16)
    images = torch.randn(128, 3, 224, 224, device=dev)
    target = torch.full((128, 1000), 1e-4, device=dev)
    trainer.start_epoch(0)
    for _ in range(4):
        trainer.step(images, target)
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        for _ in range(steps):
            trainer.step(images, target)
        torch.cuda.synchronize()
    sites = collections.defaultdict(collections.Counter)
    for ev in prof.events():
        n = ev.name
        if not any(k in n for k in ("fill_", "zero_", "copy_", "aten::mul", "aten::add", "aten::floor", "aten::rand", "aten::to", "aten::_to_copy",
                                     "aten::mean", "aten::zeros", "aten::empty")):
            continue
        if ev.device_type != torch.autograd.DeviceType.CPU or not ev.stack:
            continue
        frames = [f for f in ev.stack if "cream_amd" in f or "bench" in f or "tools/" in f][:2]
        sites[n][" <- ".join(frames) if frames else "(framework)"] += 1
    for n, c in sorted(sites.items(), key=lambda kv: -sum(kv[1].values())):
        tot = sum(c.values())
        if tot < steps:
            continue
        print(f"{n}: {tot / steps:.1f} per step")
        for s, k in c.most_common(6):
            print(f"    {k / steps:6.1f}  {s}")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 4)

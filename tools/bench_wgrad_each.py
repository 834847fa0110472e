import os, sys, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from cream_amd.autoformer import block as K
from cream_amd import _lib
DEV = "cuda:0"
def med(fn, iters=20, warmup=3):
    for _ in range(warmup): fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2] * 1e3
M = 25216
lib = _lib.load()
for name, N, Kd in (("fc2", 384, 1344), ("fc1", 1344, 384), ("proj", 384, 384), ("qkv", 1152, 384), ("fc1 E448", 1792, 448), ("fc2 E448", 448, 1792), ("qkv E448", 1344, 448)):
    dy = torch.randn(M, N, device=DEV).to(torch.bfloat16); x = torch.randn(M, Kd, device=DEV).to(torch.bfloat16)
    S = lib.cream_linear_wgrad_splits(M, N, Kd)
    t = med(lambda: K.linear_wgrad_parts(dy, x))
    tiles = ((N + 127) // 128) * ((Kd + 127) // 128)
    print(json.dumps(dict(name=name, N=N, K=Kd, S=S, tiles=tiles, grid=S * tiles, us=round(t, 1), tflops=round(2 * M * N * Kd / t / 1e6, 1))))

"""Sub-network evaluation sweep (the inner loop of AutoFormer's evolution search, evolution.py:22-290
-> supernet_engine.evaluate): inference throughput over random sub-networks of supernet-S, bf16,
per-GPU batch 128, synthetic 224^2 batches resident on the device.

    python tools/bench_subnet_eval.py [n_subnets] > gpurun_out/subnet_eval.json
"""
import json, os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cream_amd.autoformer import engine

n_sub = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda")
torch.manual_seed(0)
m = engine.build_supernet("S").to(dev)
ch = engine.SEARCH_SPACES["S"]["choices"]
g = torch.Generator(device=dev).manual_seed(1)
batches = [(torch.randn(128, 3, 224, 224, device=dev, generator=g), torch.randint(0, 1000, (128,), device=dev, generator=g))
           for _ in range(2)]
random.seed(0)
for _ in range(5):
    engine.evaluate(batches, m, choices=ch, mode="super")
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n_sub):
    r = engine.evaluate(batches, m, choices=ch, mode="super")
torch.cuda.synchronize()
dt = time.perf_counter() - t0
# the same sweep as the evolution search runs it: statistics left on the device, one synchronisation per population
random.seed(0)
torch.cuda.synchronize()
t0 = time.perf_counter()
pend = [engine.evaluate(batches, m, choices=ch, mode="super", defer=True) for _ in range(n_sub)]
rd = [p.result() for p in pend]
torch.cuda.synchronize()
dt_d = time.perf_counter() - t0
print(json.dumps(dict(workload="supernet-S sub-network evaluation sweep (eval forward, bf16, batch 128, 2 batches per sub-network)",
                      subnets=n_sub, images_per_sec=round(n_sub * 2 * 128 / dt, 1), ms_per_batch=round(dt / (n_sub * 2) * 1e3, 3),
                      subnets_per_sec=round(n_sub / dt, 2), last=dict(loss=r["loss"], params=r["params"]),
                      deferred=dict(note="statistics resolved once per population (engine.evaluate(defer=True), evolution.py)",
                                    images_per_sec=round(n_sub * 2 * 128 / dt_d, 1), ms_per_batch=round(dt_d / (n_sub * 2) * 1e3, 3),
                                    subnets_per_sec=round(n_sub / dt_d, 2)))))

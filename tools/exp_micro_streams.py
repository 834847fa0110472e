"""Experiment (timing only, not a product path): the step as two half-batches on two streams, gradients accumulated —
does the independent work of one half fill the latency gaps of the other?  (The tail / stem gradient accumulation of the two
halves is NOT ordered here: numbers only.)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch

from cream_amd.autoformer import engine

dev = torch.device("cuda")
torch.manual_seed(0)
model = engine.build_supernet("S").to(dev)
opt = engine.build_optimizer(model, batch_size=128)
tr = engine.SupernetTrainer(model, opt, engine.SEARCH_SPACES["S"]["choices"], None)
x = torch.randn(128, 3, 224, 224, device=dev)
t = torch.softmax(torch.randn(128, 1000, device=dev), -1)
tr.start_epoch(0)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def step_split(mode):
    tr.sample()
    tr.optimizer.zero_grad(set_to_none=False)
    main = torch.cuda.current_stream()
    ev0 = main.record_event()
    halves = [(x[:64], t[:64]), (x[64:], t[64:])]
    fwd_done = None
    for k, (xs, ts) in enumerate(halves):
        s = streams[k]
        s.wait_event(ev0)
        with torch.cuda.stream(s):
            if mode == "staggered" and fwd_done is not None:
                s.wait_event(fwd_done)                      # half 1 starts its forward when half 0 starts its backward
            with torch.autocast("cuda", dtype=torch.bfloat16):
                loss = engine.soft_target_cross_entropy(model(xs), ts) * 0.5
            if mode == "staggered" and k == 0:
                fwd_done = s.record_event()
            loss.backward()
    for s in streams:
        main.wait_stream(s)
    tr.optimizer.step()


def bench(fn, n=40, w=10):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


import random
for rep in range(2):
    random.seed(0)
    print("whole batch        %.3f ms/step" % bench(lambda: tr.step(x, t)), flush=True)
    random.seed(0)
    print("two halves, at once %.3f ms/step" % bench(lambda: step_split("concurrent")), flush=True)
    random.seed(0)
    print("two halves, staggered %.3f ms/step" % bench(lambda: step_split("staggered")), flush=True)

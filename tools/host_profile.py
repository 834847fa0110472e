"""Where does the HOST time of a supernet step go?  (GPU box)
Wall-clock accumulators around every C-ABI call + cProfile with single-threaded autograd (so the
hand-written backward shows up in the calling thread)."""
import cProfile, pstats, sys, os, io, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cream_amd import comm, _lib
from cream_amd.autoformer import engine, block
if '--no-side' in sys.argv:
    block.WGRAD_SIDE_STREAM = False
dev = torch.device('cuda')
torch.manual_seed(0)
model = engine.build_supernet('S', drop_path_rate=0.1).to(dev)
opt = engine.build_optimizer(model, lr=5e-4, batch_size=128, world_size=1)
red = comm.GradReducer(model)
tr = engine.SupernetTrainer(model, opt, engine.SEARCH_SPACES['S']['choices'], red, amp_dtype=torch.bfloat16)
x = torch.randn(128, 3, 224, 224, device=dev)
t = torch.softmax(torch.randn(128, 1000, device=dev), -1)
tr.start_epoch(0)
for _ in range(10):
    tr.step(x, t)
torch.cuda.synchronize()
N = 20
# ---- plain timing
t0 = time.perf_counter()
for _ in range(N):
    tr.step(x, t)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3 * (t1 - t0) / N:.2f} ms/step, total {1e3 * (t2 - t0) / N:.2f} ms/step")
# ---- C-ABI call accumulators
lib = _lib.load()
acc = collections.defaultdict(lambda: [0, 0.0])
class Timed:
    def __init__(self, name, fn): self.name, self.fn = name, fn
    def __call__(self, *a):
        s = time.perf_counter(); r = self.fn(*a); e = acc[self.name]; e[0] += 1; e[1] += time.perf_counter() - s; return r
orig = {}
for name in _lib.SIGNATURES:
    if name.startswith('cream_') and name not in ('cream_version', 'cream_build_info'):
        orig[name] = getattr(lib, name)
        setattr(lib, name, Timed(name, orig[name]))
phases = collections.defaultdict(float)
def timed_step():
    s = time.perf_counter(); tr.sample(); phases['sample+set_config'] += time.perf_counter() - s
    s = time.perf_counter(); tr.reducer.zero_grad(); tr.reducer.prepare(tr.config); phases['zero_grad'] += time.perf_counter() - s
    s = time.perf_counter()
    with torch.autocast('cuda', dtype=torch.bfloat16):
        logits = model(x); loss = engine.soft_target_cross_entropy(logits, t)
    phases['forward'] += time.perf_counter() - s
    s = time.perf_counter(); loss.backward(); phases['backward'] += time.perf_counter() - s
    s = time.perf_counter(); tr.reducer.finish(); opt.step(); phases['optimizer'] += time.perf_counter() - s
torch.autograd.set_multithreading_enabled(False)
for _ in range(3): timed_step()
torch.cuda.synchronize(); acc.clear(); phases.clear()
for _ in range(N): timed_step()
torch.cuda.synchronize()
print("phases (ms/step):", {k: round(1e3 * v / N, 3) for k, v in phases.items()})
print("C-ABI calls (per step: count, ms):")
for k, (n, s) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:34s} {n / N:6.1f} calls  {1e3 * s / N:7.3f} ms  ({1e6 * s / n:6.1f} us each)")
for name, fn in orig.items(): setattr(lib, name, fn)
pr = cProfile.Profile(); pr.enable()
for _ in range(N): tr.step(x, t)
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(22); print(s.getvalue()[:5000])

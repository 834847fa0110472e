"""Where does the HOST time of a supernet step go?  cProfile over N steps (GPU box)."""
import cProfile, pstats, sys, os, io, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cream_amd import comm
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
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    tr.step(x, t)
pr.disable()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3 * (t1 - t0) / N:.2f} ms/step (under cProfile), drain {1e3 * (t2 - t1):.1f} ms")
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28)
print(s.getvalue()[:6000])

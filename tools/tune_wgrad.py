"""Offline kernel selection for the split-K weight-gradient GEMMs at other split factors
(PyTorch TunableOp over the GEMM library's solutions; see tools/tune_gemms.py).

    python tools/tune_wgrad.py S gpurun_out/wgrad_tune.csv 16,32 [max_shapes]

Shapes: dW(out, in) = dY(M, out)^T X(M, in), M = 128 * 197 tokens, for fc1 (F x E), fc2 (E x F),
proj (E x Q) and qkv (3Q x E) over the search space.  The result lines are merged into
cream_amd/tuning/gemm_<size>_b128.csv and the per-shape best split into wgrad_split_<size>.json."""
import itertools, os, sys, time
size = sys.argv[1] if len(sys.argv) > 1 else 'S'
out = os.path.abspath(sys.argv[2] if len(sys.argv) > 2 else f'gpurun_out/wgrad_tune_{size}.csv')
splits = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else '16').split(',')]
max_shapes = int(sys.argv[4]) if len(sys.argv) > 4 else 10 ** 6
os.makedirs(os.path.dirname(out), exist_ok=True)
os.environ['PYTORCH_TUNABLEOP_ENABLED'] = '1'
os.environ['PYTORCH_TUNABLEOP_TUNING'] = '1'
os.environ['PYTORCH_TUNABLEOP_FILENAME'] = out
os.environ.setdefault('PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS', '15')
os.environ.setdefault('PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS', '2')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cream_amd.autoformer import engine
ch = engine.SEARCH_SPACES[size]['choices']
M = 128 * 197
shapes = []
for E in ch['embed_dim']:
    for R in ch['mlp_ratio']:
        F = int(E * R)
        shapes += [(F, E), (E, F)]
    for H in ch['num_heads']:
        shapes += [(E, 64 * H), (3 * 64 * H, E)]
shapes = sorted(set(shapes))[:max_shapes]
dev = torch.device('cuda')
t0 = time.time()
for i, (o, n) in enumerate(shapes):
    dy = torch.randn(M, o, device=dev).bfloat16()
    x = torch.randn(M, n, device=dev).bfloat16()
    for s in splits:
        torch.bmm(dy.view(s, M // s, -1).transpose(1, 2), x.view(s, M // s, -1))
        torch.cuda.synchronize()
    print(f"[{i + 1}/{len(shapes)}] dW {o}x{n} splits {splits}  {time.time() - t0:.0f}s", flush=True)

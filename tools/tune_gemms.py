"""Offline selection of the GEMM library's kernels for every GEMM shape of the supernet step
(PyTorch TunableOp over hipBLASLt / rocBLAS solutions).  Run on the MI355X:

    python tools/tune_gemms.py S gpurun_out/tunableop_S.csv

It walks all (embed_dim, num_heads, mlp_ratio) combinations of the search space through one
fused block forward+backward plus the stem/head GEMMs, so that every shape is seen once while
tuning is on.  The resulting CSV is committed under cream_amd/tuning/ and loaded (tuning off)
by bench.py / the trainer.
"""
import itertools, os, sys, time
size = sys.argv[1] if len(sys.argv) > 1 else 'S'
out = os.path.abspath(sys.argv[2] if len(sys.argv) > 2 else f'gpurun_out/tunableop_{size}.csv')
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 128
os.makedirs(os.path.dirname(out), exist_ok=True)
os.environ['PYTORCH_TUNABLEOP_ENABLED'] = '1'
os.environ['PYTORCH_TUNABLEOP_TUNING'] = '1'
os.environ['PYTORCH_TUNABLEOP_FILENAME'] = out
os.environ.setdefault('PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS', '40')
os.environ.setdefault('PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS', '5')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cream_amd.autoformer import engine

dev = torch.device('cuda')
space = engine.SEARCH_SPACES[size]
ch = space['choices']
model = engine.build_supernet(size, drop_path_rate=0.1, depth=2).to(dev)
model.train()
x = torch.randn(batch, 3, 224, 224, device=dev)
t = torch.softmax(torch.randn(batch, 1000, device=dev), -1)
t0 = time.time()
combos = list(itertools.product(ch['embed_dim'], ch['num_heads'], ch['mlp_ratio']))
for i, (E, H, R) in enumerate(combos):
    cfg = dict(layer_num=2, embed_dim=[E, E], num_heads=[H, H], mlp_ratio=[R, R])
    model.set_sample_config(cfg)
    model.zero_grad(set_to_none=False)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        loss = engine.soft_target_cross_entropy(model(x), t)
    loss.backward()
    torch.cuda.synchronize()
    print(f"[{i + 1}/{len(combos)}] E{E} H{H} R{R}  {time.time() - t0:.0f}s", flush=True)
print("TunableOp writes", out.replace('.csv', '0.csv'), "at exit; copy it to cream_amd/tuning/gemm_<size>_b<batch>.csv")

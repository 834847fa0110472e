"""Re-tune, over the GEMM library's OWN solutions only, the problems for which the general offline
table (tools/tune_gemms.py) picked a rocBLAS solution: the native dispatcher (csrc/gemm_lt.cpp)
can only use library-native solution indices and would otherwise fall back to the heuristic.

    python tools/tune_gemms_lt_only.py cream_amd/tuning/gemm_S_b128.csv gpurun_out/gemm_lt_only.csv

The result is committed as cream_amd/tuning/gemm_<size>_b<batch>_lt.csv and loaded AFTER the
general table (later entries override)."""
import csv, os, re, sys, time
src, out = sys.argv[1], os.path.abspath(sys.argv[2])
os.makedirs(os.path.dirname(out), exist_ok=True)
os.environ['PYTORCH_TUNABLEOP_ENABLED'] = '1'
os.environ['PYTORCH_TUNABLEOP_TUNING'] = '1'
os.environ['PYTORCH_TUNABLEOP_ROCBLAS_ENABLED'] = '0'
os.environ['PYTORCH_TUNABLEOP_HIPBLASLT_ENABLED'] = '1'
os.environ['PYTORCH_TUNABLEOP_FILENAME'] = out
os.environ.setdefault('PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS', '15')
os.environ.setdefault('PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS', '2')
import torch
dev = torch.device('cuda')
todo = [(r[0], r[1]) for r in csv.reader(open(src)) if r[0] != 'Validator' and r[2].startswith('Gemm_Rocblas_')]
t0 = time.time()
for i, (op, sig) in enumerate(todo):
    m = re.match(r'(\w)(\w)_(\d+)_(\d+)_(\d+)(?:_B_(\d+))?_ld_(\d+)_(\d+)_(\d+)', sig)
    ta, tb = m.group(1), m.group(2)
    M_, N_, K_ = int(m.group(3)), int(m.group(4)), int(m.group(5))
    batch = int(m.group(6)) if m.group(6) else 1
    lda = int(m.group(7))
    if op.startswith('GemmTunableOp') and (ta, tb) == ('n', 'n'):         # dgrad: dx(M x K) = dy(M x N) W[:N, :K]
        K, M, N = M_, N_, K_
        w = torch.randn(N, lda, device=dev).bfloat16()
        dy = torch.randn(M, N, device=dev).bfloat16()
        torch.mm(dy, w[:N, :K])
    elif op.startswith('GemmStridedBatched') and (ta, tb) == ('n', 't'):    # wgrad parts
        K, N, ms = M_, N_, K_
        dy = torch.randn(batch * ms, N, device=dev).bfloat16()
        x = torch.randn(batch * ms, K, device=dev).bfloat16()
        torch.bmm(dy.view(batch, ms, N).transpose(1, 2), x.view(batch, ms, K))
    else:
        print('skip', op, sig)
        continue
    torch.cuda.synchronize()
    print(f"[{i + 1}/{len(todo)}] {op} {sig}  {time.time() - t0:.0f}s", flush=True)

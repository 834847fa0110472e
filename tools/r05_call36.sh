#!/bin/bash
# LayerNorm kernels: the fp32 residual stream non-temporal (the bf16 output the next GEMM reads stays in L2): tests + same-call step A/B
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
cp cream_amd/libcream_amd.so /tmp/new.so
timeout 900 python -m pytest tests/test_block_gpu.py -x -q -m gpu 2>&1 | tail -2
run() { timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-host-leg 2> $OUT/ab_$1.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print('$1', d['ms_per_step'], {n: k[n]['avg_us'] for n in ('gemm_nt','gemm_nt_gelu','gemm_nt_mul','gemm_tn_wgrad','ln_fwd','ln_bwd')})"; }
for rep in 1 2 3; do
  cp gpurun_prev/libcream_amd_prev.so cream_amd/libcream_amd.so; run before_$rep
  cp /tmp/new.so cream_amd/libcream_amd.so; run after_$rep
done

#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT
for M in 2 0 3 1 2 0 3 1; do
  CREAM_GEMM_NT8=$M timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-leg --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tn8 on, nt8=$M', d['value'], d['ms_per_step'])"
done | tee $OUT/r05q_step_ab.txt

#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT
for M in 0 4 0 4 0 4; do
  CREAM_GEMM_NT8=$M timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-host-leg --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nt8=$M', d['value'], d['ms_per_step'])"
done | tee $OUT/r05s_step_ab.txt

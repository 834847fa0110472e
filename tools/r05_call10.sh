#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT
for SL in 256 128 192 256 128 192; do
  CREAM_GEMM_TN8=1 CREAM_TN8_SLOTS=$SL timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-leg --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tn8 slots=$SL', d['value'], d['ms_per_step'])"
done | tee $OUT/r05n_step_ab.txt

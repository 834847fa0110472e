#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT
for M in 0 1 0 1 0 1; do
  CREAM_FORK_MERGE=$M timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-host-leg --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fork merge=$M', d['value'], d['ms_per_step'])"
done | tee $OUT/r05y_step_ab.txt

#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT
for CFG in "1 128" "1 96" "1 64" "2 128" "2 64" "1 128" "1 96" "1 64" "2 128" "2 64"; do
  set -- $CFG
  CREAM_GEMM_TN8=$1 CREAM_TN8_SLOTS=$2 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-leg --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tn8 mode=$1 slots=$2', d['value'], d['ms_per_step'])"
done | tee $OUT/r05o_step_ab.txt

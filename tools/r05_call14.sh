#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT
for CFG in "0 256" "2 256" "2 4096" "2 192" "0 256" "2 256" "2 4096" "2 192"; do
  set -- $CFG
  CREAM_GEMM_NT8=$1 CREAM_NT8_SLOTS=$2 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-leg --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nt8=$1 slots=$2', d['value'], d['ms_per_step'])"
done | tee $OUT/r05r_step_ab.txt

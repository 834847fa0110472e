#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT
for P in 100 50 75 100 50 75; do
  CREAM_BWD_SLOTS_PCT=$P timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-leg --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bwd slots pct=$P', d['value'], d['ms_per_step'])"
done | tee $OUT/r05v_step_ab.txt

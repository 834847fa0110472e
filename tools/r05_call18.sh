#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT
for G in 2 4 8 16 32; do for T in 1024 768 512; do
  echo "G=$G thr=$T: $(CREAM_RPE_G=$G CREAM_RPE_THR=$T python tools/bench_rpe_index.py 2>/dev/null | grep rpe_index_fwd | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['dtype'][6:], round(d['GBps']), round(d['frac'], 3), end='  ')")"
done; done | tee $OUT/r05w_rpe_gather_sweep.txt

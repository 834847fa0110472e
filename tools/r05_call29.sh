#!/bin/bash
# rpe_index gather: store cache policy x launch order x G (python operator, config 4)
OUT=gpurun_out; mkdir -p $OUT
for ord in 0 1; do for pol in 0 1 2 3; do
  for cfg in "0 0" "4 1024" "8 768" "16 768" "32 512"; do set -- $cfg
    RPE_ORD=$ord RPE_POL=$pol RPE_G=$1 RPE_THR=$2 timeout 120 python tools/bench_rpe_index.py --iters 15 2>/dev/null | python -c "
import sys,json
r=[json.loads(l) for l in sys.stdin if l.startswith('{')]
f=[x for x in r if x['kernel']=='rpe_index_fwd']
print('ord $ord pol $pol G $1 thr $2:', ' '.join('%s %.0f (best %.0f) %.3f' % (x['dtype'][6:], x['GBps'], x['bytes']/x['ms_best']/1e6, x['frac']) for x in f))"
  done; done; done | tee $OUT/r05_rpe_gather_policy_sweep.txt

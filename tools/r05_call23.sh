#!/bin/bash
# runtime knobs against the 5 us behind every kernel that carries a completion signal for the side stream (same box, alternating)
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
run() { # tag, env...
  tag=$1; shift
  env "$@" timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-host-leg --no-kernel-timing 2> $OUT/knob_$tag.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', d['ms_per_step'])"
}
for rep in 1 2; do
  run base_$rep X=1
  run sss0_$rep ROC_SYSTEM_SCOPE_SIGNAL=0
  run noint_$rep HSA_ENABLE_INTERRUPT=0
  run both_$rep ROC_SYSTEM_SCOPE_SIGNAL=0 HSA_ENABLE_INTERRUPT=0
  run active_$rep ROC_ACTIVE_WAIT_TIMEOUT=1000000
done

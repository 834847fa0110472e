#!/bin/bash
# the join of the weight-gradient stream deferred to the end of the backward pass: parity tests, then a same-call step A/B
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_block_gpu.py tests/test_autoformer_gpu.py tests/test_comm_gpu.py -x -q -m gpu 2>&1 | tail -3
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-host-leg --no-kernel-timing 2> $OUT/ab_$tag.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', d['ms_per_step'], d['value'])"; }
for rep in 1 2 3 4; do
  run join_at_blocks_$rep CREAM_DEFER_JOIN=0
  run join_deferred_$rep CREAM_DEFER_JOIN=1
done

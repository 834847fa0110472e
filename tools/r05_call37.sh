#!/bin/bash
# same-call step A/B of the library in gpurun_prev/ against the one in the tree (6 runs, alternating)
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
cp cream_amd/libcream_amd.so /tmp/new.so
run() { timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-host-leg --no-kernel-timing 2> $OUT/ab_$1.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'])"; }
for rep in 1 2 3 4; do
  cp gpurun_prev/libcream_amd_prev.so cream_amd/libcream_amd.so; run before_$rep
  cp /tmp/new.so cream_amd/libcream_amd.so; run after_$rep
done
timeout 600 python -m pytest tests/test_block_gpu.py -x -q -m gpu -k "finalize or native or reproducible or block_at" 2>&1 | tail -2

#!/bin/bash
# AdamW kernel with the loads of three row groups in flight before the first store: parity test, then same-call A/B (step + the kernel's
# own duration from rocprofv3) against the previous library
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
cp cream_amd/libcream_amd.so /tmp/new.so
timeout 600 python -m pytest tests/test_block_gpu.py tests/test_autoformer_gpu.py -x -q -m gpu -k "adamw or optim or trainer or step" 2>&1 | tail -2
run() { timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-host-leg --no-kernel-timing 2> $OUT/ab_$1.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'])"; }
prof() { cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/adamw_$1 -o s -- python $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-host-leg > /dev/null 2>&1; cd $REPO
  grep -h "adamw" $(find $OUT/adamw_$1 -name '*kernel_stats.csv') | cut -d, -f1-4 | cut -c1-80; find $OUT/adamw_$1 -name '*.csv' -delete; find $OUT -name '*.db' -delete; }
for rep in 1 2 3; do
  cp gpurun_prev/libcream_amd_prev.so cream_amd/libcream_amd.so; run prev_$rep
  cp /tmp/new.so cream_amd/libcream_amd.so; run new_$rep
done
cp gpurun_prev/libcream_amd_prev.so cream_amd/libcream_amd.so; prof prev
cp /tmp/new.so cream_amd/libcream_amd.so; prof new

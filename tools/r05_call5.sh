#!/bin/bash
OUT=$(pwd)/gpurun_out; mkdir -p $OUT
REPO=$(pwd)
export TMPDIR=/tmp
cd /tmp
for M in 0 2; do
  CREAM_GEMM_NT8=$M timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r05g_prof_nt8_$M -o step -- python $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-host-leg > $OUT/r05g_prof_bench_$M.json 2> $OUT/r05g_prof_$M.err
  echo "rocprof nt8=$M exit $?"; cut -c1-200 $OUT/r05g_prof_bench_$M.json
done
find $OUT -name '*kernel_trace.csv' -path "*r05g*" -delete
find $OUT -name '*.db' -path "*r05g*" -delete

#!/bin/bash
OUT=$(pwd)/gpurun_out; mkdir -p $OUT
REPO=$(pwd)
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r05u_prof -o step -- python $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-host-leg > $OUT/r05u_prof_bench.json 2> $OUT/r05u_prof.err
echo "rocprof exit $?"; cut -c1-200 $OUT/r05u_prof_bench.json
find $OUT -name '*kernel_trace.csv' -path "*r05u*" -delete
find $OUT -name '*.db' -path "*r05u*" -delete

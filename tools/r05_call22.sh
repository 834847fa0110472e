#!/bin/bash
# where the main chain's inter-kernel gaps sit (pairs), and which host code issues the per-step fills / copies
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python tools/probes/count_fills.py > $OUT/r05_count_fills.txt 2>&1; tail -45 $OUT/r05_count_fills.txt | cut -c1-260
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/r05g_prof -o step -- python $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-host-leg > $OUT/r05g_prof_bench.json 2> $OUT/r05g_prof.err
TRACE=$(find $OUT -name '*kernel_trace.csv' -path "*r05g_prof*" | head -1)
python $REPO/tools/summarize_gaps.py $TRACE > $OUT/r05_step_gap_pairs.txt 2>&1; cat $OUT/r05_step_gap_pairs.txt | cut -c1-200
find $OUT -name '*kernel_trace.csv' -path "*r05g_prof*" -delete; find $OUT -name '*.db' -delete

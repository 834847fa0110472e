#!/bin/bash
# one visit: rocprofv3 kernel stats of the TinyCLIP config-5 leg
export TMPDIR=/tmp; REPO=$(pwd); OUT=$REPO/gpurun_out
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r06m_tc_prof -o tc -- python $REPO/tools/bench_tinyclip.py > $OUT/r06m_tc.json 2> $OUT/r06m_tc.err
cd $REPO; cat $OUT/r06m_tc.json
python tools/summarize_rocprof.py $(find $OUT/r06m_tc_prof -name '*kernel_stats.csv' | head -1) > $OUT/r06m_tinyclip_kernel_stats.md 2>&1; head -30 $OUT/r06m_tinyclip_kernel_stats.md | cut -c1-200

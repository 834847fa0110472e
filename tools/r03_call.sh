cd $GRAFT_REPO_ROOT
python tools/bench_tinyclip.py 2>&1 | tail -1 | cut -c1-200
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03_tc_prof -o tc -- python $GRAFT_REPO_ROOT/tools/bench_tinyclip.py > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/r03_tc_prof.err
cd $GRAFT_REPO_ROOT
python tools/summarize_rocprof.py $(find gpurun_out/r03_tc_prof -name '*kernel_stats.csv' | head -1) | head -34
find gpurun_out/r03_tc_prof -name '*kernel_trace.csv' -delete; find gpurun_out -name '*.db' -delete

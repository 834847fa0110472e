#!/bin/bash
# one visit: the whole GPU suite, smoke and the bench line (the short form of tools/gpu_round.sh)
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
CREAM_BENCH_EXTRA=gpurun_out/call_bench_extra.json timeout 600 python bench.py > gpurun_out/call_bench.json 2> gpurun_out/call_bench.err; cut -c1-400 gpurun_out/call_bench.json

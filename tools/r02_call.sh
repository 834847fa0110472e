#!/bin/bash
for cfg in "1024 400" "256 8" "256 16" "256 32" "256 64" "512 32" "512 64" "1024 64" "1024 128"; do set -- $cfg; echo "== NT $1 RPB $2"; CREAM_RPE_NT=$1 CREAM_RPE_RPB=$2 timeout 100 python tools/bench_rpe_index.py --iters 10 2>&1 | grep "rpe_index_fwd" | cut -c1-125; done

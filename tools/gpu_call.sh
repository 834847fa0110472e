#!/bin/bash
# one visit: config 4 as a whole model, all three ways
timeout 800 python tools/bench_deit_irpe.py 2>/dev/null | grep "^{" | tee gpurun_out/r06t_deit_irpe.jsonl | cut -c1-420

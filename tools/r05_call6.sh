#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT
for M in 0 1 0 1; do
  CREAM_GEMM_NT8=$M timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-leg --no-kernel-timing --no-wgrad-stream 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('single stream nt8=$M', d['value'], d['ms_per_step'])"
done | tee $OUT/r05h_single_stream_ab.txt
# clocks while the two-stream step runs
( CREAM_GEMM_NT8=0 python bench.py --steps 400 --warmup 5 --no-cpu-baseline --no-host-leg --no-kernel-timing > /dev/null 2>&1 & )
sleep 20
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | tr '\n' ' '; echo; sleep 0.5; done | tee $OUT/r05h_clocks.txt
wait

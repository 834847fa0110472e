#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT
(GEMM_COLD=1 tools/probes/gemm_nt_probe "fc1 fwd" "128x128 w2x2 st2 occ2 EPI_MUL"; tools/probes/gemm_nt_probe "fc1 fwd" "128x128 w2x2 st2 occ2 EPI_MUL"; tools/probes/gemm_nt_probe ragged "EPI_MUL") 2>&1 | grep -E "M=|EPI_MUL" | cut -c1-140 | tee $OUT/r05z_mul_probe.txt
timeout 600 python -m pytest tests/test_block_gpu.py -m gpu -x -q -k "gelu or block_at_bench or native_block or fused_block" 2>&1 | tail -3
for i in 1 2 3; do
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-host-leg --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('aux early', d['value'], d['ms_per_step'])"
done | tee $OUT/r05z_step.txt

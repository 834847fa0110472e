#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT
for v in p1_false p0_false p0_true p2_true p1_true; do tools/probes/nt8_trace_probe_$v 25216 448 1792; done > $OUT/r05b_trace.txt 2>&1
grep -E "variant|cycles per K-tile|phase" $OUT/r05b_trace.txt
GEMM_COLD=1 timeout 600 tools/probes/gemm_nt_probe "fc" > $OUT/r05b_probe_cold.txt 2>&1
grep -c WRONG $OUT/r05b_probe_cold.txt; grep -E "library|nt8|256x256 w2x4 st2 occ1 EPI" $OUT/r05b_probe_cold.txt | cut -c1-100

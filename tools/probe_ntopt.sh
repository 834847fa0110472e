#!/bin/bash
# phase stamps of the first tile of every workgroup (100 MHz ticks): where a tile's time goes
P=tools/probes/gemm_nt_probe
export GEMM_COLD=1 GEMM_PHASES=1
for S in "fc1 fwd   E384"; do timeout 200 $P "$S" "128x128"; done
for S in "fc2 fwd   E384"; do timeout 200 $P "$S" "256x192"; done

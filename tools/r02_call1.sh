#!/bin/bash
# round 2, GPU visit 1: own-GEMM probe vs library, host-side profile of the step, baseline bench line
OUT=gpurun_out; mkdir -p $OUT
timeout 300 tools/probes/gemm_nt_probe > $OUT/r02a_gemm_nt_probe.txt 2>&1; echo "probe exit $?"
tail -120 $OUT/r02a_gemm_nt_probe.txt
timeout 300 python tools/host_profile.py > $OUT/r02a_host_profile.txt 2>&1; echo "hostprof exit $?"
head -60 $OUT/r02a_host_profile.txt
timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline > $OUT/r02a_bench.json 2> $OUT/r02a_bench.err; echo "bench exit $?"
cat $OUT/r02a_bench.json

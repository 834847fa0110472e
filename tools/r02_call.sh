#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT
timeout 1200 python -m pytest tests -q -m gpu > $OUT/r02g_pytest_gpu.log 2>&1; echo "pytest exit $?"
tail -15 $OUT/r02g_pytest_gpu.log
grep -E "^(FAILED|ERROR)" $OUT/r02g_pytest_gpu.log | head -30
timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline > $OUT/r02g_bench.json 2> $OUT/r02g_bench.err; echo "bench exit $?"
cat $OUT/r02g_bench.json; tail -5 $OUT/r02g_bench.err
timeout 300 python tools/host_profile.py > $OUT/r02g_host_profile.txt 2>&1; echo "hostprof exit $?"
head -30 $OUT/r02g_host_profile.txt

#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu -s > $OUT/r02i_pytest_gpu.log 2>&1; echo "pytest exit $?"
grep -E "^\[|passed|failed" $OUT/r02i_pytest_gpu.log | tail -30
grep -E "^(FAILED|ERROR)" $OUT/r02i_pytest_gpu.log | head -30
grep -B5 -A25 "Error\b" $OUT/r02i_pytest_gpu.log | head -150
timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline > $OUT/r02i_bench.json 2> $OUT/r02i_bench.err; echo "bench exit $?"
cut -c1-400 $OUT/r02i_bench.json; tail -3 $OUT/r02i_bench.err

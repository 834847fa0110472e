#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT
for M in 0 2 1 0 2 1; do
  CREAM_GEMM_NT8=$M timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-leg --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nt8=$M', d['value'], d['ms_per_step'])"
done | tee $OUT/r05e_step_ab.txt
timeout 900 python -m pytest tests/test_block_gpu.py tests/test_autoformer_gpu.py -m gpu -x -q > $OUT/r05e_pytest.log 2>&1
echo "pytest exit $?"; tail -3 $OUT/r05e_pytest.log

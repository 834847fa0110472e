#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_block_gpu.py tests/test_autoformer_gpu.py -m gpu -x -q > $OUT/r05l_pytest.log 2>&1
echo "pytest exit $?"; tail -3 $OUT/r05l_pytest.log
for M in 0 1 2 0 1 2; do
  CREAM_GEMM_TN8=$M timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-leg --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tn8=$M', d['value'], d['ms_per_step'])"
done | tee $OUT/r05l_step_ab.txt
for M in 0 1; do
  CREAM_GEMM_TN8=$M timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-leg --no-kernel-timing --no-wgrad-stream 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('single stream tn8=$M', d['value'], d['ms_per_step'])"
done | tee -a $OUT/r05l_step_ab.txt

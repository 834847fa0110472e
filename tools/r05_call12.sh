#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT
GEMM_COLD=1 tools/probes/gemm_nt_probe wgrad > $OUT/r05p_tn_probe.txt 2>&1
echo "WRONG lines: $(grep -c WRONG $OUT/r05p_tn_probe.txt)"; grep -E "tn8" $OUT/r05p_tn_probe.txt | grep "bias1" | cut -c1-200 | head -24
timeout 900 python -m pytest tests/test_block_gpu.py tests/test_autoformer_gpu.py -m gpu -x -q > $OUT/r05p_pytest.log 2>&1
echo "pytest exit $?"; tail -3 $OUT/r05p_pytest.log
for M in 0 2 0 2; do
  CREAM_GEMM_TN8=$M timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-leg --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tn8=$M', d['value'], d['ms_per_step'])"
done | tee $OUT/r05p_step_ab.txt

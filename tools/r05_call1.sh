#!/bin/bash
# round 5, first GPU visit: the phase-interleaved NT kernel — correctness + race screen, cold probe next to the library,
# phase stamps, the block / model parity tests on it, step A/B
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
( timeout 120 tools/probes/gemm_nt_probe ragged; timeout 120 tools/probes/gemm_nt_probe segment ) > $OUT/r05a_probe_small.txt 2>&1
echo "small probe exit $?"; grep -c WRONG $OUT/r05a_probe_small.txt; grep nt8 $OUT/r05a_probe_small.txt | cut -c1-150
GEMM_COLD=1 timeout 600 tools/probes/gemm_nt_probe E > $OUT/r05a_probe_cold.txt 2>&1
echo "cold probe exit $?"; grep -c WRONG $OUT/r05a_probe_cold.txt; grep -E "library|nt8|256x256 w2x4 st2 occ1 EPI_BIAS |128x128 w2x2 st2 occ2 EPI_BIAS " $OUT/r05a_probe_cold.txt | cut -c1-140
for S in "fc1 fwd   E384" "fc2 fwd   E384" "fc2 fwd   E448" "proj fwd  E384"; do
  GEMM_COLD=1 GEMM_PHASES=1 timeout 120 tools/probes/gemm_nt_probe_prof "$S" nt8
done > $OUT/r05a_probe_phases.txt 2>&1
grep phases $OUT/r05a_probe_phases.txt | cut -c1-260
timeout 900 python -m pytest tests/test_block_gpu.py tests/test_autoformer_gpu.py -m gpu -x -q > $OUT/r05a_pytest.log 2>&1
echo "pytest exit $?"; tail -5 $OUT/r05a_pytest.log
for M in 0 1 0 1; do
  CREAM_GEMM_NT8=$M timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-leg --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nt8=$M', d['value'], d['ms_per_step'])"
done | tee $OUT/r05a_step_ab.txt

#!/bin/bash
# same-call A/B of the NT kernel policy on the TinyCLIP config-5 leg (student width 512, teacher 768)
for E in "X=1" "CREAM_GEMM_NT8=1" "CREAM_GEMM_NT8=2" "CREAM_GEMM_NT256=1"; do for r in 1 2; do
  echo "[$E] $(env $E timeout 300 python tools/bench_tinyclip.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['loss'])")"
done; done

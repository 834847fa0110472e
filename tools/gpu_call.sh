#!/bin/bash
timeout 1200 python -m pytest tests/test_block_gpu.py tests/test_deit_native_gpu.py tests/test_tinyclip_model.py -m gpu -x -q 2>&1 | tail -3
for E in "X=1" "CREAM_GEMM_NT8=1"; do for r in 1 2; do
    echo "[$E] $(env $E DEIT_ONLY=k1 timeout 200 python tools/bench_deit_irpe.py 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['loss'])")"
done; done
timeout 300 python tools/bench_tinyclip.py 2>/dev/null | cut -c1-200

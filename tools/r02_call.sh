#!/bin/bash
timeout 900 python -m pytest tests/test_irpe_fused_gpu.py tests/test_irpe_gpu.py -q 2>&1 | tail -3 | cut -c1-300
timeout 300 python tools/bench_irpe_attention.py 2>/dev/null | grep bfloat16 | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print(d['workload'][-10:], d['ms_per_fwd_bwd'], {k:(v['avg_us'], v['TFLOPs']) for k,v in d['kernels'].items()})"

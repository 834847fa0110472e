#!/bin/bash
timeout 1200 python -m pytest tests/test_attn_gpu.py tests/test_block_gpu.py tests/test_autoformer_gpu.py -q -x 2>&1 | tail -3 | cut -c1-300
timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'], {k:(v['avg_us'],v['tflops']) for k,v in d['roofline']['kernels'].items()})"

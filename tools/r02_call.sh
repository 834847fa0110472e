#!/bin/bash
timeout 900 python -m pytest tests/test_block_gpu.py tests/test_autoformer_gpu.py -q -x -s 2>&1 | grep -v "^$" | tail -8 | cut -c1-300
for rep in 1 2; do
timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-kernel-timing 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'])"
done

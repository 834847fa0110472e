#!/bin/bash
timeout 900 python -m pytest tests/test_block_gpu.py -q -x -s -k padded 2>&1 | grep "padded\|passed\|failed\|Error" | cut -c1-400
timeout 300 python bench.py --supernet T --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('T', d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'])"

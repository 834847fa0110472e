#!/bin/bash
for rep in 1 2 3; do for V in 0 1; do
CREAM_NATIVE_ENDS=$V timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-kernel-timing 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('native_ends $V', d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'])"
done; done

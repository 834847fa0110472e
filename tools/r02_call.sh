#!/bin/bash
for rep in 1 2 3; do for S in 512 256 384; do
CREAM_WGRAD_SLOTS=$S timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-kernel-timing 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('slots $S', d['value'], d['ms_per_step'])"
done; done

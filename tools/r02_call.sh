#!/bin/bash
for rep in 1 2; do
timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-kernel-timing 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default(now 1)', d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'])"
HIP_FORCE_DEV_KERNARG=0 timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-kernel-timing 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('forced 0', d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'])"
done
for E in "AMD_DIRECT_DISPATCH=0" "GPU_MAX_HW_QUEUES=2" "HIP_LAUNCH_BLOCKING=0"; do
env $E timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-kernel-timing 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$E', d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'])"
done

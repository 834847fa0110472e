#!/bin/bash
python - <<'PY'
import torch, ctypes
print("torch priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else None)
for p in (-2,-1,0,1,2):
    try:
        s=torch.cuda.Stream(priority=p); print("prio",p,"->",s.priority)
    except Exception as e: print("prio",p,"ERR",e)
PY
for rep in 1 2; do for V in 0 1 -1; do
CREAM_SIDE_PRIORITY=$V timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-kernel-timing 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('side_prio $V', d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'])"
done; done

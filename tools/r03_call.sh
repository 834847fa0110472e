cd $GRAFT_REPO_ROOT
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())"
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-host-leg"
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"])'
for rep in 1 2 3; do
$B 2>/dev/null | python -c "$P" "default            "
CREAM_MAIN_PRIORITY=-1 $B 2>/dev/null | python -c "$P" "main high          "
CREAM_SIDE_PRIORITY=1 $B 2>/dev/null | python -c "$P" "side low(+1)       "
CREAM_SIDE_PRIORITY=-1 $B 2>/dev/null | python -c "$P" "side HIGH          "
CREAM_MAIN_PRIORITY=0 $B 2>/dev/null | python -c "$P" "main on own stream0"
done

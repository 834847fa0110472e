cd $GRAFT_REPO_ROOT
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-host-leg"
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"])'
for rep in 1 2 3; do
$B 2>/dev/null | python -c "$P" "default            "
CREAM_NT_NARROW=1 $B 2>/dev/null | python -c "$P" "all NT 128x64 occ3 "
done

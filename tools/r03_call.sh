cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for C in 16 32; do
CREAM_WGRAD_SCAP=$C python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-host-leg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cap $C', d['value'], d['ms_per_step'])"
done; done

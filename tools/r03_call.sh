cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
CREAM_ATTN_OLD_FWD=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-host-leg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('old fwd', d['value'], d['ms_per_step'])"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-host-leg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('whole-KV fwd', d['value'], d['ms_per_step'])"
done

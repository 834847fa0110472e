cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_block_gpu.py tests/test_autoformer_gpu.py -m gpu -x -q -s 2>&1 | grep -E "grouped|passed|failed|Error|assert" | tail -12
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-host-leg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'])"

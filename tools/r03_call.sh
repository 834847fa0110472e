cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
CREAM_NT_PIPE=1 timeout 600 python -m pytest tests/test_block_gpu.py -x -q -m gpu -k "native_linear or qkv_segment or fused_gelu or hidden_width or block_at_bench or native_block_sequencing or patch_embedding" 2>&1 | tail -5
for rep in 1 2 3; do
for pipe in 0 1 3; do
CREAM_NT_PIPE=$pipe python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-host-leg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pipe=$pipe', d['value'], d['ms_per_step'])"
done
done

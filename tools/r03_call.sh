cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_block_gpu.py tests/test_attn_rpe2d_gpu.py tests/test_autoformer_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python tools/bench_subnet_eval.py 2>/dev/null | cut -c1-300

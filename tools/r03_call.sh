cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_block_gpu.py tests/test_tinyclip_model.py -x -q -m gpu -k "without_backward or native_tower or fused_gelu" 2>&1 | tail -3
timeout 300 python tools/bench_subnet_eval.py 2>/dev/null | cut -c1-400
timeout 300 python tools/bench_tinyclip.py 2>/dev/null | tail -2 | cut -c1-400

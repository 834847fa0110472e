cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_irpe_fused_gpu.py tests/test_irpe_gpu.py tests/test_minivit.py tests/test_tinyclip_model.py -m gpu -x -q 2>&1 | tail -4
python tools/bench_irpe_attention.py 2>/dev/null | cut -c1-400

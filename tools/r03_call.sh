cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_irpe_fused_gpu.py tests/test_tinyclip_model.py tests/test_irpe_gpu.py -m gpu -x -q -s 2>&1 | grep -E "^\.*\[causal|^\.*\[tinyclip|passed|failed|Error|assert" | tail -12
CREAM_TINYCLIP_NATIVE=0 python tools/bench_tinyclip.py 2>/dev/null | cut -c1-200
python tools/bench_tinyclip.py 2>&1 | tail -2 | cut -c1-200

cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_block_gpu.py -m gpu -x -q -s -k "soft_target_ce or native_head" 2>&1 | grep -E "^\.*\[|passed|failed|Error|assert" | tail

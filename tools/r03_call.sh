cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_tinyclip_model.py -m gpu -x -q -s 2>&1 | grep -E "^\.*\[|passed|failed|Error|assert" | tail -8

#!/bin/bash
timeout 900 python -m pytest tests/test_deit_native_gpu.py tests/test_tinyclip_model.py tests/test_tinyclip_loss.py -m gpu -x -q -s 2>&1 | grep -E "deit native|passed|failed|Error|error|assert" | cut -c1-500 | tail -14

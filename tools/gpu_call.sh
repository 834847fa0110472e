#!/bin/bash
# one visit: the native DeiT block stack — parity tests, TinyCLIP tests (same node), then config 4 as a whole model
timeout 900 python -m pytest tests/test_deit_native_gpu.py tests/test_tinyclip_model.py tests/test_tinyclip_loss.py tests/test_irpe_gpu.py -m gpu -x -q -s 2>&1 | grep -E "deit native|passed|failed|Error|error|assert" | cut -c1-700 | tail -20


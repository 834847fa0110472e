#!/bin/bash
# one visit: gradient finalisation in overwrite mode — its test, the stacks that use it, the two model-level legs
timeout 900 python -m pytest tests/test_block_gpu.py tests/test_deit_native_gpu.py tests/test_tinyclip_model.py tests/test_tinyclip_loss.py tests/test_autoformer_gpu.py -m gpu -x -q 2>&1 | tail -3
DEIT_ONLY=k1 timeout 300 python tools/bench_deit_irpe.py 2>/dev/null | grep "^{" | cut -c1-330
timeout 300 python tools/bench_tinyclip.py 2>/dev/null | cut -c1-300

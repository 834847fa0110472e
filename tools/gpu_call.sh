#!/bin/bash
# one visit: the attention-family parity tests, then the config-4 layer and the TinyCLIP leg
timeout 1500 python -m pytest tests/test_irpe_fused_gpu.py tests/test_irpe_gpu.py tests/test_minivit.py tests/test_detr_attention.py tests/test_tinyclip_model.py tests/test_tinyclip_loss.py -m gpu -x -q 2>&1 | tail -4
timeout 300 python tools/bench_irpe_attention.py 2>/dev/null | tee gpurun_out/r06o_irpe_attention.jsonl | cut -c1-330
timeout 300 python tools/bench_tinyclip.py 2>/dev/null | tee gpurun_out/r06o_tinyclip.json
bash tools/irpe_round.sh r06o notest | tail -24

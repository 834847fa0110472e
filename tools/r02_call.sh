#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_irpe_fused_gpu.py -q -s 2>&1 | tail -30 | cut -c1-400

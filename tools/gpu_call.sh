#!/bin/bash
# one visit: the whole GPU suite, then the config-4 layer benchmark
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
timeout 300 python tools/bench_irpe_attention.py 2>/dev/null | tee gpurun_out/r06m_irpe_attention.jsonl | cut -c1-420

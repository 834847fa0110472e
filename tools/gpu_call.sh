#!/bin/bash
# config 4 as a whole model with DeiT's stochastic depth (drop_path 0.1): own kernels against framework blocks
for N in k1 k0; do DEIT_DROP_PATH=0.1 DEIT_ONLY=$N timeout 300 python tools/bench_deit_irpe.py 2>/dev/null | grep "^{" | cut -c60-460; done | tee gpurun_out/r06w_deit_drop_path.txt

#!/bin/bash
# one visit: the new parity tests, then the same-call A/B of the step under the new switches
timeout 900 python -m pytest tests/test_attn_gpu.py tests/test_block_gpu.py -m gpu -x -q -k "role_split or nt_epilogue or gelu_table or kernel_variants" 2>&1 | tail -15
bash tools/ab_step.sh 3 "CREAM_ATTN_BWD1=1 CREAM_GEMM_NTOPT=0" "CREAM_ATTN_BWD1=2 CREAM_GEMM_NTOPT=0" "CREAM_ATTN_BWD1=1 CREAM_GEMM_NTOPT=3" "CREAM_ATTN_BWD1=2 CREAM_GEMM_NTOPT=3" "CREAM_ATTN_BWD1=2 CREAM_GEMM_NTOPT=1"

#!/bin/bash
# one visit: the fused iRPE attention at long sequences (L = 1025, 2026: beyond what the suite covers)
cd tests && PYTHONPATH=.. timeout 600 python - <<'PY'
import torch, test_irpe_fused_gpu as T
for case in [("qkv", True, 1025, "product"), ("k", False, 1025, "product"), ("qkv", True, 2026, "product"), ("kv", True, 2026, "euc", "ctx", 14.0), ("qk", False, 1025, "quant", "bias")]:
    try:
        T.test_fused_irpe_attention_matches_restatement(case); print("ok", case)
    except Exception as e:
        print("FAIL", case, repr(e)[:300])
PY

#!/bin/bash
# same-call A/B with the in-step kernel timing on: step, gemm_tn8 and attention in-step durations per switch setting
R=$1; shift
for r in $(seq 1 $R); do
  for E in "$@"; do
    env $E CREAM_BENCH_EXTRA=/dev/null timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-host-leg > /tmp/ab.json 2> /tmp/ab.err
    echo "round $r  [$E]  rc $?  $(python -c "import json;d=json.load(open('/tmp/ab.json'));r=d['roofline'];print(d['ms_per_step'], 'ms  tn8', r['avg_us'], 'us frac', r['frac'], ' attn', r['attention_us'])" 2>/dev/null)"
  done
done

#!/bin/bash
# same-call A/B of the training step under environment switches: bash tools/ab_step.sh <rounds> "VAR=a VAR2=b" "VAR=c" ...
R=$1; shift
for r in $(seq 1 $R); do
  for E in "$@"; do
    env $E timeout 200 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-timing --no-host-leg > /tmp/ab.json 2> /tmp/ab.err
    echo "round $r  [$E]  rc $?  $(python -c "import json;d=json.load(open('/tmp/ab.json'));print(d['ms_per_step'], 'ms', d['value'], 'img/s')" 2>/dev/null)"
  done
done

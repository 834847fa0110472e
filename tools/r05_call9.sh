#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT
for MODE in "" "--no-wgrad-stream"; do
  python bench.py --steps 1500 --warmup 5 --no-cpu-baseline --no-host-leg --no-kernel-timing $MODE > $OUT/r05m_bench_long.json 2>/dev/null &
  BP=$!
  sleep 14
  echo "== mode '$MODE'"
  for i in $(seq 1 12); do /opt/rocm/bin/rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E "sclk|Power \(W\)|GPU use" | sed 's/.*: //' | tr '\n' ' '; echo; sleep 0.4; done
  wait $BP
  python -c "import json; d=json.loads(open('$OUT/r05m_bench_long.json').read().strip().splitlines()[-1]); print('ms/step', d['ms_per_step'])"
done 2>&1 | tee $OUT/r05m_clocks.txt

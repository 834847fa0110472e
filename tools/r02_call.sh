#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT
timeout 300 tools/probes/gemm_nt_probe "fc1 fwd   E384" > $OUT/r02q_probe.txt 2>&1; grep -v "^tr16\|fast erf" $OUT/r02q_probe.txt | cut -c1-150
timeout 300 tools/probes/gemm_nt_probe "fc2 fwd   E384" 2>&1 | grep -v "^tr16\|fast erf" | cut -c1-150
timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'], {k:(v['avg_us'],v['tflops']) for k,v in d['roofline']['kernels'].items()})"

#!/bin/bash
# one visit: the default bench command, timed
( time CREAM_BENCH_EXTRA=gpurun_out/r06u_bench_extra.json timeout 900 python bench.py > gpurun_out/r06u_bench.json 2> gpurun_out/r06u_bench.err ) 2>&1 | grep real
cut -c1-200 gpurun_out/r06u_bench.json; tail -3 gpurun_out/r06u_bench.err
python -c "
import json; e=json.load(open('gpurun_out/r06u_bench_extra.json')); print(e['irpe_config4'].get('model')); print(e['tinyclip_config5']['ms_per_step'])"

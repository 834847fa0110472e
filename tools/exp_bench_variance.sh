rocm-smi --showperflevel --showpower 2>&1 | grep -E "Perf|Power" | head -3
for i in 1 2; do python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('nocpu', d['ms_per_step'], d['roofline']['kernels'])"; done
python bench.py --steps 40 --warmup 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cpu  ', d['ms_per_step'], d['roofline']['kernels'], d['cpu_baseline'])"
python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('nocpu', d['ms_per_step'], d['roofline']['kernels'])"
python bench.py --steps 200 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('nocpu200', d['ms_per_step'], d['roofline']['kernels'])"
rocm-smi --showclocks 2>&1 | grep -E "sclk|mclk" | head -3

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
/opt/rocm/bin/rocm-smi --showclocks --showpower --showmaxpower 2>/dev/null | grep -v "^$" | head -30
bash tools/sample_clocks.sh gpurun_out/r03_clocks_step.txt python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-host-leg --no-kernel-timing 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
grep -c sclk gpurun_out/r03_clocks_step.txt

#!/bin/bash
# One GPU-box visit for the fused iRPE kernels: parity tests, per-subset kernel times, per-kernel rocprofv3 durations.
# usage: bash tools/irpe_round.sh <tag> [notest]
TAG=${1:-irpe}
OUT=$(pwd)/gpurun_out; mkdir -p $OUT; REPO=$(pwd)
export TMPDIR=/tmp
if [ "$2" != "notest" ]; then
timeout 900 python -m pytest tests/test_irpe_fused_gpu.py tests/test_irpe_gpu.py tests/test_minivit.py tests/test_detr_attention.py tests/test_tinyclip_model.py -m gpu -x -q 2>&1 | tail -8
fi
timeout 300 python tools/bench_irpe_terms.py 2>&1 | tee $OUT/${TAG}_terms.jsonl
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/${TAG}_prof -o t -- python $REPO/tools/bench_irpe_terms.py > /dev/null 2> $OUT/${TAG}_prof.err
cd $REPO
python - <<PY | tee $OUT/${TAG}_kernels.txt
import csv, collections, glob
f = glob.glob("$OUT/${TAG}_prof/**/*kernel_trace.csv", recursive=True)
d = collections.defaultdict(list); meta = {}
for r in csv.DictReader(open(f[0])):
    n = r["Kernel_Name"]
    if "irpe_" not in n: continue
    n = n.replace("void (anonymous namespace)::", "").split("(")[0]
    d[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    meta[n] = (r["LDS_Block_Size"], r["VGPR_Count"])
for n in sorted(d):
    v = sorted(d[n]); print(f"{n:60s} n={len(v):3d} median={v[len(v)//2]:8.1f} us  lds={meta[n][0]} vgpr={meta[n][1]}")
PY

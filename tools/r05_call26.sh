#!/bin/bash
# the waits of the main chain for the side stream (exposed tail of the weight gradients at the end of the backward)
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/r05h_prof -o step -- python $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-host-leg > $OUT/r05h_prof_bench.json 2> $OUT/r05h_prof.err
TRACE=$(find $OUT -name '*kernel_trace.csv' -path "*r05h_prof*" | head -1)
python $REPO/tools/summarize_gaps.py $TRACE > $OUT/r05_step_waits.txt 2>&1; grep -A14 "waits of" $OUT/r05_step_waits.txt | cut -c1-200
# one step's tail in detail: the last 30 kernels before the second-to-last adamw, both queues
python - "$TRACE" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
key = "Queue_Id" if "Queue_Id" in rows[0] else "Stream_Id"
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r[key], r["Kernel_Name"]) for r in rows))
idx = [i for i, r in enumerate(rows) if "adamw" in r[3]]
i = idx[-3]
t0 = rows[i - 30][0]
for s, e, q, n in rows[i - 30:i + 2]:
    n = n.replace("cream::gemm::", "").replace("(anonymous namespace)::", "")
    print(f"q{q} {1e-3 * (s - t0):9.1f} .. {1e-3 * (e - t0):9.1f} us  {n[:70]}")
PY
find $OUT -name '*kernel_trace.csv' -path "*r05h_prof*" -delete; find $OUT -name '*.db' -delete

#!/bin/bash
OUT=$(pwd)/gpurun_out; mkdir -p $OUT
REPO=$(pwd)
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r05x_prof -o step -- python $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-host-leg > $OUT/r05x_prof_bench.json 2> $OUT/r05x_prof.err
TRACE=$(find $OUT -name '*kernel_trace.csv' -path "*r05x_prof*" | head -1)
python $REPO/tools/summarize_gaps.py $TRACE > $OUT/r05x_step_gaps.txt 2>&1; cat $OUT/r05x_step_gaps.txt
# keep a compact per-kernel timeline of one step for inspection: name, queue, start, end (ns)
python - "$TRACE" > $OUT/r05x_timeline_one_step.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last 700 kernels ~ two steps
t0 = int(rows[-700]["Start_Timestamp"])
for r in rows[-700:]:
    print(r.get("Queue_Id"), (int(r["Start_Timestamp"]) - t0) // 100 / 10, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) // 100 / 10, r["Kernel_Name"][:60], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", ""))
PY
find $OUT -name '*kernel_trace.csv' -path "*r05x*" -delete
find $OUT -name '*.db' -path "*r05x*" -delete

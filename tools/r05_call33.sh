#!/bin/bash
# HBM traffic of the GELU / x gelu' / plain NT kernels ALONE (probe, cold operands) — is the +30 % over algorithmic seen in the step the kernels' own?
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  GEMM_COLD=1 timeout 300 rocprofv3 --kernel-trace --pmc $C -d $OUT/ntpmc_$C -o pmc --output-format csv -- $REPO/tools/probes/gemm_nt_probe "fc1 fwd   E384" "128x128 w2x2 st2 occ2" > /dev/null 2> $OUT/ntpmc_$C.err
done
python - <<'PY'
import csv, glob, collections, os
out = os.environ.get("OUT", "/root/repo/gpurun_out")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{out}/ntpmc_{C}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == C:
                acc[r["Kernel_Name"][:90]][C].append(float(r["Counter_Value"]))
for k, v in acc.items():
    if "gemm_nt_kernel" not in k: continue
    f = sum(v["FETCH_SIZE"]) / max(1, len(v["FETCH_SIZE"])); w = sum(v["WRITE_SIZE"]) / max(1, len(v["WRITE_SIZE"]))
    # FETCH_SIZE in units of 64 B (KB per the guide's correction x2?) -- print raw and the guide's conversion (KB -> bytes, x2 for gfx950)
    print(k, "launches", len(v["FETCH_SIZE"]), "FETCH raw %.0f WRITE raw %.0f" % (f, w), "-> read %.1f MB (2 x FETCH_SIZE KiB) write %.1f MB" % (f * 2 / 1024, w / 1024))
PY

"""Per-kernel means of raw rocprofv3 --pmc counters as a markdown table (+ a few ratios).
usage: python tools/summarize_pmc_table.py '<glob of pmc dirs>' '<kernel regex>' > profiles/x.md"""
import collections
import csv
import glob
import os
import re
import sys


def main():
    dirs, pat = sys.argv[1], re.compile(sys.argv[2])
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in sorted(glob.glob(dirs)):
        for f in glob.glob(os.path.join(d, "*counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                m = pat.search(r["Kernel_Name"])
                if m:
                    vals[m.group(0)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    counters = sorted({c for d in vals.values() for c in d})
    print(f"source: rocprofv3 --kernel-trace --pmc <group> (one pass per group), dirs `{dirs}`; means per launch\n")
    print("| kernel | launches | " + " | ".join(counters) + " |")
    print("|---|---:|" + "---:|" * len(counters))
    for k, d in sorted(vals.items()):
        n = max(len(v) for v in d.values())
        print(f"| `{k}` | {n} | " + " | ".join(f"{sum(d[c]) / len(d[c]):.4g}" if c in d else "" for c in counters) + " |")
    print("\nderived:\n")
    for k, d in sorted(vals.items()):
        m = {c: sum(v) / len(v) for c, v in d.items()}
        out = []
        if "FETCH_SIZE" in m:
            out.append(f"HBM read {m['FETCH_SIZE'] * 2 / 1024:.1f} MB (2 x FETCH_SIZE KiB, gfx950 rule)")
        if "WRITE_SIZE" in m:
            out.append(f"HBM write {m['WRITE_SIZE'] / 1024:.1f} MB")
        if m.get("SQ_WAVE_CYCLES"):
            w = m["SQ_WAVE_CYCLES"]
            for c, name in (("SQ_WAIT_ANY", "parked (s_waitcnt / barrier)"), ("SQ_WAIT_INST_ANY", "issue stalls"),
                            ("SQ_ACTIVE_INST_ANY", "issuing")):
                if c in m:
                    out.append(f"{name} {100 * m[c] / w:.0f} % of wave cycles")
        if m.get("SQ_LDS_IDX_ACTIVE") and "SQ_LDS_BANK_CONFLICT" in m:
            out.append(f"LDS bank-conflict cycles {100 * m['SQ_LDS_BANK_CONFLICT'] / m['SQ_LDS_IDX_ACTIVE']:.0f} % of LDS-active cycles")
        print(f"* `{k}`: " + "; ".join(out))


if __name__ == "__main__":
    main()

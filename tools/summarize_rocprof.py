"""Condense a rocprofv3 `*_kernel_stats.csv` into a short table for profiles/.
usage: python tools/summarize_rocprof.py gpurun_out/prof_x/x_kernel_stats.csv [steps] > profiles/rNN_x.md"""
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"(Custom_)?Cijk_(\w+?)_(\w+?)_.*?MT(\d+x\d+x\d+)", name)
    if m:
        return f"hipBLASLt GEMM {m.group(2)}_{m.group(3)} MT{m.group(4)}"
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"at::native::", "", name)
    return name[:110]


def main():
    path = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else None
    rows = list(csv.DictReader(open(path)))
    total = sum(int(r["TotalDurationNs"]) for r in rows)
    print(f"source: {path}")
    print(f"total kernel time: {total / 1e6:.2f} ms" + (f" over {steps} steps = {total / 1e6 / steps:.2f} ms/step" if steps else ""))
    print()
    print("| % | calls | avg us | total ms | kernel |")
    print("|---:|---:|---:|---:|---|")
    agg = {}
    for r in rows:
        k = short(r["Name"])
        a = agg.setdefault(k, [0, 0])
        a[0] += int(r["Calls"])
        a[1] += int(r["TotalDurationNs"])
    for k, (calls, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f"| {100 * ns / total:.2f} | {calls} | {ns / calls / 1e3:.1f} | {ns / 1e6:.2f} | `{k}` |")


if __name__ == "__main__":
    main()

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


def category(k):
    if k.startswith("hipBLASLt GEMM") or "Cijk_" in k:
        return "GEMM library (hipBLASLt)"
    if "gemm_nt_kernel" in k or "gemm_tn_kernel" in k or "gemm_nt8_kernel" in k or "gemm_tn8_kernel" in k:
        return "GEMM (csrc/gemm_mfma.hpp, gemm_nt8.hpp, gemm_tn8.hpp: hand-written MFMA)"
    if "adamw_mirror" in k:
        return "optimizer + bf16 operand refresh (csrc/optim.hip)"
    if "attn_rpe2d" in k:
        return "attention (csrc/attn_rpe2d.hip)"
    if re.search(r"(ln_fwd|ln_bwd|gelu_|residual_add|scale_cast|colsum|grad_finalize|rpe_)", k):
        return "HBM passes (csrc/block_ops.hip, rpe_index.hip)"
    if "FusedAdam" in k or "multi_tensor_apply" in k:
        return "optimizer + bf16 operand refresh (multi-tensor)"
    return "framework elementwise / reductions / copies / fills"


def read_rows(path):
    """rocprofv3 --stats output: either <x>_kernel_stats.csv or the rocpd database <x>_results.db."""
    if path.endswith(".db"):
        import sqlite3
        c = sqlite3.connect(path)
        q = "select name, count(*), sum(duration) from kernels group by name"
        return [{"Name": n, "Calls": k, "TotalDurationNs": d} for n, k, d in c.execute(q)]
    return list(csv.DictReader(open(path)))


def main():
    path = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else None
    rows = read_rows(path)
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
    cats = {}
    for k, (calls, ns) in agg.items():
        c = cats.setdefault(category(k), [0, 0])
        c[0] += calls
        c[1] += ns
    print()
    print("| category | calls | total ms |" + (" ms/step |" if steps else ""))
    print("|---|---:|---:|" + ("---:|" if steps else ""))
    for k, (calls, ns) in sorted(cats.items(), key=lambda kv: -kv[1][1]):
        print(f"| {k} | {calls} | {ns / 1e6:.2f} |" + (f" {ns / 1e6 / steps:.2f} |" if steps else ""))


if __name__ == "__main__":
    main()

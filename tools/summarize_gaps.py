"""Idle time between consecutive kernels of each HSA queue in a rocprofv3 kernel trace (csv): how much of the
step is spent BETWEEN kernels of the main chain (launch / barrier-packet / cache-maintenance gaps) and how much of the
wall time has any kernel running at all.
usage: python tools/summarize_gaps.py <..._kernel_trace.csv> [skip_first_n_kernels]"""
import csv
import sys
from collections import defaultdict


def short(name):
    name = name.replace("cream::gemm::", "").replace("void ", "")
    for a, b in (("vectorized_elementwise_kernel<4, FillFunctor<float>", "torch fill<float>"), ("__amd_rocclr_", "rocclr_")):
        if name.startswith(a):
            return b
    cut = name.find("(")
    return (name[:cut] if cut > 0 else name)[:56]


def main(path, skip=0):
    rows = list(csv.DictReader(open(path)))
    key = "Queue_Id" if "Queue_Id" in rows[0] else "Stream_Id"
    rows = [(r[key], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
    rows.sort(key=lambda r: r[1])
    rows = rows[skip:]
    t0, t1 = rows[0][1], max(r[2] for r in rows)
    per = defaultdict(list)
    for q, s, e, n in rows:
        per[q].append((s, e, n))
    print(f"window {1e-6 * (t1 - t0):.2f} ms, {len(rows)} kernels, {len(per)} queues")
    for q, ks in sorted(per.items(), key=lambda kv: -len(kv[1])):
        busy = sum(e - s for s, e, _ in ks)
        gaps = [max(0, ks[i + 1][0] - ks[i][1]) for i in range(len(ks) - 1)]
        small = [g for g in gaps if g < 20000]            # < 20 us: back-to-back launches, not a wait for another stream / the host
        hist = defaultdict(int)
        for g in small:
            hist[min(g // 1000, 10)] += 1
        print(f"queue {q}: {len(ks)} kernels, busy {1e-6 * busy:.2f} ms, gaps < 20 us: n = {len(small)}, total {1e-6 * sum(small):.3f} ms, "
              f"median {sorted(small)[len(small) // 2] / 1e3 if small else 0:.2f} us; larger gaps total {1e-6 * (sum(gaps) - sum(small)):.2f} ms")
        print("   gap histogram (us: count): " + ", ".join(f"{k}-{k + 1}: {v}" for k, v in sorted(hist.items())))
        # which (previous -> next) kernel pairs the gaps of 2-20 us sit between
        pairs = defaultdict(lambda: [0, 0])
        for i, g in enumerate(gaps):
            if 2000 <= g < 20000:
                k = (short(ks[i][2]), short(ks[i + 1][2]))
                pairs[k][0] += 1
                pairs[k][1] += g
        for (a, b), (n, tot) in sorted(pairs.items(), key=lambda kv: -kv[1][1])[:24]:
            print(f"      {n:5d} x {tot / n / 1e3:5.2f} us = {tot / 1e6:6.3f} ms   {a}  ->  {b}")
        # the waits of this queue for the other one (or for the host): gaps of 20 us .. 3 ms by pair
        waits = defaultdict(lambda: [0, 0])
        for i, g in enumerate(gaps):
            if 20000 <= g < 3000000:
                k = (short(ks[i][2]), short(ks[i + 1][2]))
                waits[k][0] += 1
                waits[k][1] += g
        if waits:
            print("   waits of 20 us .. 3 ms:")
        for (a, b), (n, tot) in sorted(waits.items(), key=lambda kv: -kv[1][1])[:12]:
            print(f"      {n:5d} x {tot / n / 1e3:7.1f} us = {tot / 1e6:6.3f} ms   {a}  ->  {b}")
    # union of busy intervals over all queues
    ev = sorted((s, e) for _, s, e, _ in rows)
    cov, cur_s, cur_e = 0, ev[0][0], ev[0][1]
    for s, e in ev[1:]:
        if s > cur_e:
            cov += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    cov += cur_e - cur_s
    print(f"some kernel running: {1e-6 * cov:.2f} ms of {1e-6 * (t1 - t0):.2f} ms ({100.0 * cov / (t1 - t0):.1f} %)")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 0)

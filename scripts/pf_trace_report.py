#!/usr/bin/env python
"""Fast request against slow request, side by side, from a rocprofv3 csv trace of scripts/pf_trace.py (kernel + memory-copy + HIP
runtime API): which kernel stretches, which gap opens, which host call blocks.

    python scripts/pf_trace_report.py gpurun_out/pf [marker-kernel-substring]

A request ends with the one launch of the persistent sampler (marker, default "dit_sample"); kernels are grouped between
consecutive marker ends.  For the last requests the report gives: device span, sum of kernel durations, the largest gaps with the
kernels on either side, per-kernel totals of the median-length request next to the longest one, the memory copies inside the
longest one and every HIP API call above 1 ms in its window.
"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name: str, n: int = 64) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name[:n]


def load(dirname, suffix):
    files = glob.glob(os.path.join(dirname, "**", f"*{suffix}.csv"), recursive=True)
    rows = []
    for f in files:
        with open(f, newline="") as fh:
            rows += list(csv.DictReader(fh))
    return rows


def col(row, *names):
    for n in names:
        if n in row:
            return row[n]
    raise KeyError(f"none of {names} in {list(row)}")


def main(dirname, marker="dit_sample"):
    ker = load(dirname, "kernel_trace")
    if not ker:
        print(f"# no *kernel_trace.csv under {dirname}")
        return
    K = sorted(((col(r, "Kernel_Name"), int(col(r, "Start_Timestamp")), int(col(r, "End_Timestamp")),
                 col(r, "Stream_Id", "Queue_Id")) for r in ker), key=lambda x: x[1])
    cop = [(col(r, "Direction"), int(col(r, "Start_Timestamp")), int(col(r, "End_Timestamp"))) for r in load(dirname, "memory_copy_trace")]
    api = [(col(r, "Function"), int(col(r, "Start_Timestamp")), int(col(r, "End_Timestamp"))) for r in load(dirname, "hip_api_trace")]
    ends = [k[2] for k in K if marker in k[0]]
    print(f"# {len(K)} kernels, {len(cop)} copies, {len(api)} HIP API calls, {len(ends)} requests (marker '{marker}')")
    reqs = []
    for i in range(1, len(ends)):
        ks = [k for k in K if ends[i - 1] < k[1] and k[2] <= ends[i]]
        if ks:
            reqs.append((ends[i - 1], ends[i], ks))
    reqs = reqs[-18:]
    print(f"\n{'req':>4s} {'period_ms':>10s} {'dev_span_ms':>12s} {'sum_kernels_ms':>15s} {'n':>5s} {'largest_gap_ms':>15s}  gap between")
    stats = []
    for i, (p0, p1, ks) in enumerate(reqs):
        span = ks[-1][2] - ks[0][1]
        tot = sum(k[2] - k[1] for k in ks)
        gaps = sorted(((ks[j + 1][1] - max(x[2] for x in ks[: j + 1]), j) for j in range(len(ks) - 1)), reverse=True)
        g, j = gaps[0]
        stats.append((p1 - p0, span, tot, gaps))
        print(f"{i:4d} {(p1 - p0) / 1e6:10.2f} {span / 1e6:12.2f} {tot / 1e6:15.2f} {len(ks):5d} {g / 1e6:15.3f}  {short(ks[j][0], 40)} -> {short(ks[j + 1][0], 40)}")
    order = sorted(range(len(reqs)), key=lambda i: stats[i][0])
    fast, slow = order[len(order) // 2], order[-1]
    print(f"\n# median-period request {fast} against the longest {slow}: per-kernel totals (ms), calls")
    tabs = []
    for i in (fast, slow):
        by = defaultdict(lambda: [0, 0])
        for k in reqs[i][2]:
            e = by[short(k[0])]
            e[0] += 1
            e[1] += k[2] - k[1]
        tabs.append(by)
    names = sorted(set(tabs[0]) | set(tabs[1]), key=lambda n: -(tabs[1].get(n, [0, 0])[1]))
    print(f"{'kernel':64s} {'n_fast':>6s} {'ms_fast':>9s} {'n_slow':>6s} {'ms_slow':>9s} {'ratio':>6s}")
    for n in names[:30]:
        a, b = tabs[0].get(n, [0, 0]), tabs[1].get(n, [0, 0])
        print(f"{n:64s} {a[0]:6d} {a[1] / 1e6:9.3f} {b[0]:6d} {b[1] / 1e6:9.3f} {(b[1] / a[1]) if a[1] else float('nan'):6.2f}")
    for tag, i in (("median", fast), ("longest", slow)):
        p0, p1, ks = reqs[i]
        print(f"\n# {tag} request {i}: the 8 largest gaps (ms) and what surrounds them; times relative to the previous request's end")
        for g, j in stats[i][3][:8]:
            print(f"  gap {g / 1e6:8.3f} at +{(ks[j][2] - p0) / 1e6:8.3f} ms: {short(ks[j][0], 44)} [{ks[j][3]}] -> {short(ks[j + 1][0], 44)} [{ks[j + 1][3]}]")
        print(f"  first kernel at +{(ks[0][1] - p0) / 1e6:.3f} ms: {short(ks[0][0], 50)}; marker kernel {(ks[-1][2] - ks[-1][1]) / 1e6:.3f} ms")
        cc = [c for c in cop if p0 <= c[1] <= p1]
        print(f"  {len(cc)} copies: " + ", ".join(f"{c[0]} +{(c[1] - p0) / 1e6:.2f} ({(c[2] - c[1]) / 1e3:.0f} us)" for c in cc[:12]))
        aa = sorted((a for a in api if p0 <= a[1] <= p1 and a[2] - a[1] > 1_000_000), key=lambda a: a[1])
        print("  HIP API calls > 1 ms: " + "; ".join(f"{a[0]} +{(a[1] - p0) / 1e6:.2f} for {(a[2] - a[1]) / 1e6:.2f} ms" for a in aa[:12]))
        byf = defaultdict(lambda: [0, 0])
        for a in api:
            if p0 <= a[1] <= p1:
                e = byf[a[0]]
                e[0] += 1
                e[1] += a[2] - a[1]
        print("  HIP API totals: " + ", ".join(f"{n} x{c} {t / 1e6:.2f} ms" for n, (c, t) in sorted(byf.items(), key=lambda x: -x[1][1])[:8]))


if __name__ == "__main__":
    main(*sys.argv[1:3])

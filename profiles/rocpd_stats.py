#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average duration.

    rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o NAME -- python bench.py ...
    python profiles/rocpd_stats.py gpurun_out/prof/NAME_results.db > profiles/NAME_kernel_stats.txt
    python profiles/rocpd_stats.py --per-step A_results.db 3 B_results.db 9      # exact per-step table from two traces
    python profiles/rocpd_stats.py --pmc PMC_results.db [kernel-substring,...]    # hardware counters per kernel
    python profiles/rocpd_stats.py --names NAME_results.db                         # full (untruncated) kernel names
"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"void ", "", name)
    return name[:110]


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       "from kernels group by name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    print(f"# {path}: {sum(r[1] for r in rows)} dispatches, {total/1e6:.1f} ms GPU kernel time; columns {cols[:4]}...")
    print(f"{'kernel':110s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'%':>6s}")
    for n, c, s, a, mn, mx in rows[:45]:
        print(f"{short(n):110s} {c:7d} {s/1e6:10.2f} {a/1e3:10.1f} {mn/1e3:9.1f} {mx/1e3:9.1f} {100*s/total:6.2f}")


def names(path):
    """full kernel names (Tensile encodes macro tile, wave tiling, prefetch depths ... in them) with calls and average time"""
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select name, count(*), avg(end-start) from kernels group by name order by sum(end-start) desc").fetchall()
    for n, c, a in rows[:40]:
        print(f"{c:6d} calls  {a/1e3:9.1f} us avg  {n}")


def per_step(path_a, steps_a, path_b, steps_b):
    """two traces of the same command with steps_a < steps_b timed+warm-up steps: (B - A) / (steps_b - steps_a) is the exact
    per-step kernel time, free of model construction / initialisation launches"""
    tabs = []
    for p in (path_a, path_b):
        cur = sqlite3.connect(p).cursor()
        tabs.append({n: (c, s) for n, c, s in cur.execute("select name, count(*), sum(end-start) from kernels group by name")})
    a, b = tabs
    d = float(steps_b - steps_a)
    rows = sorted(((n, (b[n][0] - a.get(n, (0, 0))[0]) / d, (b[n][1] - a.get(n, (0, 0))[1]) / d) for n in b), key=lambda r: -r[2])
    total = sum(r[2] for r in rows)
    print(f"# per training step = ({path_b} - {path_a}) / {int(d)}: {sum(r[1] for r in rows):.0f} launches, {total/1e6:.2f} ms GPU kernel time")
    print(f"{'kernel':110s} {'calls/step':>10s} {'ms/step':>10s} {'avg_us':>10s} {'%':>6s}")
    for n, c, s in rows[:60]:
        if c > 0:
            print(f"{short(n):110s} {c:10.1f} {s/1e6:10.3f} {s/1e3/c:10.1f} {100*s/total:6.2f}")


def pmc(path, only=None):
    """per-kernel sums of the hardware counters of a `rocprofv3 --pmc ... --kernel-trace` run (view counters_collection);
    `only`: comma-separated kernel-name substrings -> every counter of those kernels, not just the 60 largest rows"""
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    kcol = "kernel_name" if "kernel_name" in cols else ("name" if "name" in cols else cols[0])
    ccol = "counter_name" if "counter_name" in cols else "pmc_name"
    vcol = "value" if "value" in cols else "counter_value"
    print(f"# {path}: counters_collection columns {cols}")
    rows = cur.execute(f"select {kcol}, {ccol}, count(distinct dispatch_id), sum({vcol}) from counters_collection "
                       f"group by {kcol}, {ccol} order by 4 desc").fetchall()
    print(f"{'kernel':90s} {'counter':28s} {'dispatches':>10s} {'sum':>16s} {'per_dispatch':>16s}")
    if only:
        rows = [r for r in rows if any(o in r[0] for o in only.split(","))]
    else:
        rows = rows[:60]
    for n, c, k, v in rows:
        print(f"{short(n)[:90]:90s} {c:28s} {k:10d} {v:16.4g} {v / max(k, 1):16.4g}")


if __name__ == "__main__":
    if "--per-step" in sys.argv:
        r = [a for a in sys.argv[1:] if a != "--per-step"]
        per_step(r[0], int(r[1]), r[2], int(r[3]))
    elif "--names" in sys.argv:
        names([a for a in sys.argv[1:] if a != "--names"][0])
    elif "--pmc" in sys.argv:
        rest = [a for a in sys.argv[1:] if a != "--pmc"]
        pmc(rest[0], rest[1] if len(rest) > 1 else None)
    else:
        main(sys.argv[1])

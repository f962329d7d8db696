#!/usr/bin/env python
"""Where does a training step wait?  From a rocprofv3 --kernel-trace database (rocpd sqlite): take ONE steady-state step (between the
ends of the last two adamw_k launches), lay its kernels out per HIP stream / hardware queue and report
  * per stream: launches, busy time (union of its kernels), share of the step;
  * the stream that carries the step (largest busy time): every gap between consecutive kernels — launch latency, a wait on
    another stream, the host — summed by the kernel that PRECEDES the gap, and the largest gaps one by one;
  * kernels of the other streams that run while the main stream is idle (what it is probably waiting for).

    rocprofv3 --kernel-trace -d gpurun_out/prof -o tl -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-latency \\
        --no-secondary --no-recipe
    python scripts/step_timeline.py gpurun_out/prof/tl_results.db > profiles/r05_step_timeline.txt
"""
import re
import sqlite3
import sys
from collections import defaultdict


def short(name: str, n: int = 70) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name[:n]


def union(iv):
    iv = sorted(iv)
    tot, cs, ce = 0, None, None
    for s, e in iv:
        if cs is None:
            cs, ce = s, e
        elif s <= ce:
            ce = max(ce, e)
        else:
            tot += ce - cs
            cs, ce = s, e
    if cs is not None:
        tot += ce - cs
    return tot


def main(path, marker="adamw_k"):
    cur = sqlite3.connect(path).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    lane = next((c for c in ("stream_id", "queue_id", "stream", "queue", "tid") if c in cols), None)
    print(f"# {path}: kernels view columns: {cols}; lane column: {lane}")
    sel = f"select name, start, end, {lane or '0'} from kernels order by start"
    rows = cur.execute(sel).fetchall()
    marks = [r for r in rows if marker in r[0]]
    if len(marks) < 2:
        print(f"# fewer than two {marker} launches: nothing to cut a step from")
        return
    t0, t1 = marks[-2][2], marks[-1][2]
    step = [r for r in rows if r[1] >= t0 and r[2] <= t1]
    dur = t1 - t0
    print(f"# one step = end of {marker} #{len(marks) - 1} -> end of {marker} #{len(marks)}: {dur / 1e6:.2f} ms, {len(step)} launches")
    lanes = defaultdict(list)
    for r in step:
        lanes[r[3]].append(r)
    busy_all = union([(r[1], r[2]) for r in step])
    print(f"# device busy (union over all streams) {busy_all / 1e6:.2f} ms = {100 * busy_all / dur:.1f} % of the step; "
          f"sum of kernel durations {sum(r[2] - r[1] for r in step) / 1e6:.2f} ms")
    print(f"{'lane':>12s} {'launches':>9s} {'busy_ms':>9s} {'share':>7s}  largest kernels")
    order = sorted(lanes, key=lambda k: -union([(r[1], r[2]) for r in lanes[k]]))
    for k in order:
        b = union([(r[1], r[2]) for r in lanes[k]])
        by = defaultdict(float)
        for r in lanes[k]:
            by[short(r[0], 40)] += r[2] - r[1]
        top = ", ".join(f"{n} {v / 1e6:.1f}" for n, v in sorted(by.items(), key=lambda x: -x[1])[:3])
        print(f"{str(k):>12s} {len(lanes[k]):9d} {b / 1e6:9.2f} {100 * b / dur:6.1f}%  {top}")
    main_lane = order[0]
    m = sorted(lanes[main_lane], key=lambda r: r[1])
    print(f"\n# main stream {main_lane}, by kernel (what the step's length is made of)")
    byk = defaultdict(lambda: [0, 0.0])
    for r in m:
        e = byk[short(r[0], 90)]
        e[0] += 1
        e[1] += r[2] - r[1]
    print(f"{'kernel':90s} {'calls':>6s} {'ms':>9s} {'avg_us':>8s} {'%step':>6s}")
    for n, (c, v) in sorted(byk.items(), key=lambda x: -x[1][1])[:45]:
        print(f"{n:90s} {c:6d} {v / 1e6:9.3f} {v / 1e3 / c:8.1f} {100 * v / dur:6.2f}")
    others = sorted((r for k in order[1:] for r in lanes[k]), key=lambda r: r[1])
    gaps = []
    prev_end = t0
    prev_name = f"<{marker} of the previous step>"
    for r in m:
        if r[1] > prev_end:
            gaps.append((r[1] - prev_end, prev_end, r[1], prev_name, r[0]))
        if r[2] > prev_end:
            prev_end, prev_name = r[2], r[0]
    tot_gap = sum(g[0] for g in gaps)
    print(f"\n# main stream {main_lane}: {len(gaps)} gaps, {tot_gap / 1e6:.2f} ms idle in total ({100 * tot_gap / dur:.1f} % of the step); "
          f"gaps < 5 us: {sum(g[0] for g in gaps if g[0] < 5000) / 1e6:.2f} ms in {sum(1 for g in gaps if g[0] < 5000)}, "
          f">= 5 us: {sum(g[0] for g in gaps if g[0] >= 5000) / 1e6:.2f} ms in {sum(1 for g in gaps if g[0] >= 5000)}")
    by_prev = defaultdict(lambda: [0, 0.0])
    for g in gaps:
        e = by_prev[short(g[3])]
        e[0] += 1
        e[1] += g[0]
    print(f"{'idle after kernel':70s} {'gaps':>6s} {'idle_ms':>9s} {'avg_us':>8s}")
    for n, (c, v) in sorted(by_prev.items(), key=lambda x: -x[1][1])[:25]:
        print(f"{n:70s} {c:6d} {v / 1e6:9.3f} {v / 1e3 / c:8.1f}")
    # where in the step the queue runs dry: idle time and launches per 10 ms window of the step
    W = 10_000_000
    nwin = int(dur // W) + 1
    idle_w, launch_w = [0.0] * nwin, [0] * nwin
    for g in gaps:
        a, b = g[1] - t0, g[2] - t0
        w = int(a // W)
        while a < b and w < nwin:
            e = min(b, (w + 1) * W)
            idle_w[w] += e - a
            a, w = e, w + 1
    for r in m:
        launch_w[min(nwin - 1, int((r[1] - t0) // W))] += 1
    print("\n# per 10 ms window of the step: launches on the main stream, idle ms (a window with many launches and much idle = host-bound)")
    for w in range(nwin):
        bar = "#" * int(round(idle_w[w] / W * 40))
        print(f"  +{w * 10:4d} ms  launches {launch_w[w]:5d}  idle {idle_w[w] / 1e6:6.2f} ms  {bar}")
    print("\n# the 25 largest gaps: length, what ran before / after on the main stream, what the OTHER streams ran meanwhile")
    for g in sorted(gaps, key=lambda x: -x[0])[:25]:
        mean = [o for o in others if o[2] > g[1] and o[1] < g[2]]
        by = defaultdict(float)
        for o in mean:
            by[short(o[0], 36)] += min(o[2], g[2]) - max(o[1], g[1])
        oth = ", ".join(f"{n} {v / 1e3:.0f}us" for n, v in sorted(by.items(), key=lambda x: -x[1])[:3]) or "-"
        print(f"{g[0] / 1e3:8.1f} us at +{(g[1] - t0) / 1e6:7.2f} ms  after {short(g[3], 40):40s} before {short(g[4], 40):40s} | {oth}")


if __name__ == "__main__":
    main(*sys.argv[1:])

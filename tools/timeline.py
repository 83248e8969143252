#!/usr/bin/env python
"""Per-stream occupancy of one training step from a rocprofv3 kernel trace (rocpd SQLite): for every HIP stream /
queue the busy time, and for the whole step the time with 0 / 1 / >= 2 kernels in flight.
usage: python tools/timeline.py <results.db> [skip_steps] """
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    rows = cur.execute(f"select name, start, end, {qcol or '0'} from kernels order by start").fetchall()
    # find step boundaries: adam_kernel launches (3 per step); use the last full step
    adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[0]]
    if len(adam) < 6:
        print("not enough steps in the trace")
        return
    lo, hi = adam[-4] + 1, adam[-1] + 1          # kernels after the 3rd-last step's last adam .. last adam
    step = rows[lo:hi]
    t0, t1 = step[0][1], max(r[2] for r in step)
    print(f"step span {(t1 - t0) / 1e6:.3f} ms, {len(step)} kernels")
    per = {}
    for n, s, e, q in step:
        per[q] = per.get(q, 0) + (e - s)
    for q, v in sorted(per.items(), key=lambda kv: -kv[1]):
        print(f"  queue {q}: busy {v / 1e6:.3f} ms")
    ev = sorted([(s, 1) for _, s, e, _ in step] + [(e, -1) for _, s, e, _ in step])
    depth, last, hist = 0, t0, {}
    for t, d in ev:
        hist[min(depth, 2)] = hist.get(min(depth, 2), 0) + (t - last)
        depth += d
        last = t
    for k in sorted(hist):
        print(f"  {k if k < 2 else '>=2'} kernels in flight: {hist[k] / 1e6:.3f} ms")
    # the longest single-kernel-in-flight stretches: what runs alone
    alone = {}
    depth, last, cur_names = 0, t0, []
    active = []
    for n, s, e, q in step:
        pass
    import heapq
    ends = []
    tl = sorted(step, key=lambda r: r[1])
    i = 0
    points = sorted(set([r[1] for r in step] + [r[2] for r in step]))
    act = []
    j = 0
    for a, b in zip(points[:-1], points[1:]):
        while j < len(tl) and tl[j][1] <= a:
            act.append(tl[j])
            j += 1
        act = [r for r in act if r[2] > a]
        if len(act) == 1:
            nm = act[0][0].split("(")[0][-50:]
            alone[nm] = alone.get(nm, 0) + (b - a)
    print("  time spent ALONE on the GPU, by kernel:")
    for n, v in sorted(alone.items(), key=lambda kv: -kv[1])[:18]:
        print(f"    {v / 1e6:7.3f} ms  {n}")


if __name__ == "__main__":
    main()

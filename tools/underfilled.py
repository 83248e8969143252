#!/usr/bin/env python
"""Where is the chip UNDER-FILLED inside one replayed training step?  From a rocprofv3 kernel trace (rocpd SQLite) of the default
schedule: every interval between two launch boundaries is classified by the LARGEST grid among the kernels in flight; the time with
nothing or only small grids (< `small` workgroups) in flight is the latency-bound share of the step.  Lists the longest runs of
consecutive under-filled intervals with the kernels that make them up -- the chains worth fusing or moving beside big kernels.
usage: python tools/underfilled.py <results.db> [small=128] [runs=25]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:]+)(<[^(]*>)?\(", name)
    return ((m.group(1) + (m.group(2) or "")) if m else name)[:60]


def main():
    db = sqlite3.connect(sys.argv[1])
    small = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    nruns = int(sys.argv[3]) if len(sys.argv) > 3 else 25
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    gx = next((c for c in ("grid_size_x", "grid_x") if c in cols), None)
    wx = next((c for c in ("workgroup_size_x", "workgroup_x") if c in cols), None)
    gy = next((c for c in ("grid_size_y", "grid_y") if c in cols), None)
    wy = next((c for c in ("workgroup_size_y", "workgroup_y") if c in cols), None)
    sel = f"name, start, end, {gx or '0'}, {wx or '1'}, {gy or '1'}, {wy or '1'}"
    rows = cur.execute(f"select {sel} from kernels order by start").fetchall()
    adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[0]]
    if len(adam) < 6:
        print("not enough steps in the trace")
        return
    step = rows[adam[-4] + 1:adam[-1] + 1]
    t0, t1 = step[0][1], max(r[2] for r in step)

    def wgs(r):                                      # rocprofv3 reports the grid in work-ITEMS
        return max(1, int(r[3]) // max(1, int(r[4]))) * max(1, int(r[5]) // max(1, int(r[6])))
    pts = sorted(set([r[1] for r in step] + [r[2] for r in step]))
    tl = sorted(step, key=lambda r: r[1])
    act, j = [], 0
    under, runs, cur_run = 0.0, [], None
    for a, b in zip(pts[:-1], pts[1:]):
        while j < len(tl) and tl[j][1] <= a:
            act.append(tl[j]); j += 1
        act = [r for r in act if r[2] > a]
        big = max((wgs(r) for r in act), default=0)
        if big < small:
            under += b - a
            if cur_run is None:
                cur_run = [a, b, {}]
            cur_run[1] = b
            for r in act:
                k = short(r[0])
                cur_run[2][k] = cur_run[2].get(k, 0) + 1
        elif cur_run is not None:
            runs.append(cur_run); cur_run = None
    if cur_run is not None:
        runs.append(cur_run)
    print(f"step span {(t1 - t0) / 1e6:.3f} ms, {len(step)} kernels; under-filled (largest grid in flight < {small} workgroups, or idle): "
          f"{under / 1e6:.3f} ms in {len(runs)} runs")
    for a, b, names in sorted(runs, key=lambda r: r[0] - r[1])[:nruns]:
        print(f"  {(b - a) / 1e3:8.1f} us at +{(a - t0) / 1e6:7.3f} ms: " + ", ".join(sorted(names)))


if __name__ == "__main__":
    main()

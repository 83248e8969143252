#!/usr/bin/env python
"""Which kernels of the step pay for a resident neighbour?  Two rocprofv3 kernel traces of tools/cu_contention.py (k = 0 and k = 1
neighbour workgroups, serial schedule), grouped by (kernel, workgroups in the grid): average duration alone vs beside the neighbour.
usage: python tools/contention_kernels.py <alone.db> <beside.db> [num_CUs]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:]+)(<[^(]*>)?\(", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:70]


def load(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    gx = [c for c in ("grid_x", "grid_size_x") if c in cols]
    wx = [c for c in ("workgroup_x", "workgroup_size_x") if c in cols]
    if gx and wx:
        g = gx[0].replace("_x", "")
        w = wx[0].replace("_x", "")
        q = (f"select name, end - start, ({g}_x * {g}_y * {g}_z) / ({w}_x * {w}_y * {w}_z) from kernels")
    else:
        q = "select name, end - start, 0 from kernels"
    agg = {}
    for name, d, wgs in cur.execute(q):
        a = agg.setdefault((short(name), int(wgs)), [0, 0])
        a[0] += 1
        a[1] += d
    return agg


def main():
    a0, a1 = load(sys.argv[1]), load(sys.argv[2])
    ncu = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    rows = []
    for k in a0:
        if k in a1 and "load_path_probe" not in k[0]:
            n0, d0 = a0[k]
            n1, d1 = a1[k]
            rows.append((k, n0, d0 / n0 / 1e3, d1 / n1 / 1e3))
    t0 = sum(n * u0 for _, n, u0, _ in rows)
    t1 = sum(n * u1 for _, n, _, u1 in rows)
    print(f"sum over matched (kernel, grid) groups, weighted by the calls of the undisturbed trace: {t0 / 1e3:.3f} ms alone, {t1 / 1e3:.3f} ms beside "
          f"one resident workgroup (+{100 * (t1 / t0 - 1):.1f} %)")
    print(f"{'kernel':58s} {'wgs':>7s} {'wgs/CU':>7s} {'calls':>6s} {'alone_us':>9s} {'beside_us':>9s} {'ratio':>6s} {'delta_ms':>9s}")
    for k, n, u0, u1 in sorted(rows, key=lambda r: -(r[3] - r[2]) * r[1])[:60]:
        print(f"{k[0][:58]:58s} {k[1]:7d} {k[1] / ncu:7.2f} {n:6d} {u0:9.1f} {u1:9.1f} {u1 / u0:6.2f} {(u1 - u0) * n / 1e3:9.3f}")
    # by grid class
    cls = {}
    for k, n, u0, u1 in rows:
        r = k[1] / ncu
        c = "< 0.5 round" if r < 0.5 else "0.5-1 round" if r <= 1.0 else "1-2 rounds" if r <= 2.0 else "2-4" if r <= 4 else "> 4"
        x = cls.setdefault(c, [0.0, 0.0])
        x[0] += n * u0
        x[1] += n * u1
    print("by workgroups per CU in the grid:")
    for c, (x0, x1) in cls.items():
        print(f"  {c:12s} {x0 / 1e3:9.3f} ms -> {x1 / 1e3:9.3f} ms  (+{100 * (x1 / max(x0, 1e-9) - 1):.1f} %)")


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Launches that cannot fill the chip: from a rocprofv3 kernel trace (rocpd SQLite), every kernel name with its launch count,
mean duration and mean workgroup count, filtered to launches of fewer than 256 workgroups that run longer than 15 us -- a
one-thread dot product hid in such a launch for three rounds.
usage: python tools/small_grid_hunt.py <results.db> [steps]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    gx = [c for c in cols if c.lower() in ("grid_x", "grid_size_x", "grid_size")]
    wx = [c for c in cols if c.lower() in ("workgroup_x", "workgroup_size_x", "workgroup_size")]
    if not gx or not wx:
        print("columns:", cols)
        return
    gy = gx[0].replace("x", "y") if gx[0].endswith("x") else None
    gz = gx[0].replace("x", "z") if gx[0].endswith("x") else None
    wy = wx[0].replace("x", "y") if wx[0].endswith("x") else None
    wz = wx[0].replace("x", "z") if wx[0].endswith("x") else None
    g = f"({gx[0]}" + (f" * {gy} * {gz}" if gy in cols else "") + ")"
    w = f"({wx[0]}" + (f" * {wy} * {wz}" if wy in cols else "") + ")"
    rows = cur.execute(f"select name, count(*), avg(end - start), avg(1.0 * {g} / {w}), min(1.0 * {g} / {w}), sum(end - start) "
                       f"from kernels group by name, cast({g} / {w} as int)").fetchall()
    out = [r for r in rows if r[3] < 256 and r[2] > 15e3]
    print(f"{'kernel':70s} {'n/step':>7s} {'avg us':>8s} {'wgs':>7s} {'ms/step':>8s}")
    for n, c, a, wg, wmin, tot in sorted(out, key=lambda r: -r[5]):
        print(f"{n[:70]:70s} {c / steps:7.1f} {a / 1e3:8.1f} {wg:7.0f} {tot / 1e6 / steps:8.3f}")


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Every launch of the kernels whose name contains a pattern, in launch order, from a rocprofv3 (rocpd SQLite) kernel trace:
duration, grid, and the name of the launch in front of it (what ran before decides whether its operands are still in L2 / MALL).
usage: python tools/kernel_launches.py <results.db> <pattern> [max_rows]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    pat = sys.argv[2]
    lim = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    gcols = [c for c in ("grid_x", "grid_size_x", "workgroup_x", "workgroup_size_x") if c in cols]
    rows = cur.execute(f"select name, start, end{''.join(', ' + c for c in gcols)} from kernels order by start").fetchall()
    n = 0
    for i, r in enumerate(rows):
        if pat in r[0]:
            prev = rows[i - 1][0].split("(")[0][-40:] if i else ""
            print(f"{(r[2] - r[1]) / 1e3:9.1f} us  {' '.join(f'{c}={v}' for c, v in zip(gcols, r[3:]))}   after {prev}")
            n += 1
            if n >= lim:
                break


if __name__ == "__main__":
    main()

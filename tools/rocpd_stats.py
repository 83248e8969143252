#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd SQLite) kernel trace: per-kernel calls / total / average / share.
usage: python tools/rocpd_stats.py <results.db> [steps_in_trace]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:]+)(<[^(]*>)?\(", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:90]


def main():
    db = sqlite3.connect(sys.argv[1])
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    rows = cur.execute("select name, end - start from kernels").fetchall() if "name" in cols else []
    agg = {}
    for name, d in rows:
        a = agg.setdefault(short(name), [0, 0])
        a[0] += 1
        a[1] += d
    tot = sum(a[1] for a in agg.values())
    print(f"{'kernel':90s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'%':>6s}  (per step: /{steps:g})")
    for k, (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{k[:90]:90s} {n:7d} {d / 1e6 / steps:10.3f} {d / n / 1e3:9.1f} {100 * d / tot:6.2f}")
    print(f"{'TOTAL':90s} {sum(a[0] for a in agg.values()):7d} {tot / 1e6 / steps:10.3f}")


if __name__ == "__main__":
    main()

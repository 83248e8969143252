#!/usr/bin/env python
"""Aggregate one rocprofv3 --pmc pass of SQ counters per kernel: share of wave cycles spent issuing (ACTIVE), parked in
s_waitcnt / barriers (WAIT_ANY), issue-stalled (WAIT_INST_ANY), and VALU instructions per wave.
usage: python tools/pmc_sq.py <dir with *counter_collection.csv>"""
import collections
import csv
import glob
import os
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:]+)(<[^(]*>)?\(", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:80]


def main():
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.Counter()
    for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            acc[k]["_waves"] += 0
            key = (r["Dispatch_Id"], k)
            if key not in seen:
                seen.add(key)
                calls[k] += 1
                acc[k]["_threads"] += float(r["Grid_Size"])
    print(f"{'kernel':58s} {'calls':>5s} {'waveMcyc':>9s} {'act%':>5s} {'wait%':>5s} {'stall%':>6s} {'VALU/wave':>9s} {'LDS/wave':>8s}")
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
        wc = v.get("SQ_WAVE_CYCLES", 0.0)
        if wc <= 0:
            continue
        waves = v["_threads"] / 64.0
        print(f"{k[:58]:58s} {calls[k]:5d} {wc * 4 / 1e6:9.1f} {100 * v.get('SQ_ACTIVE_INST_ANY', 0) / wc:5.1f} "
              f"{100 * v.get('SQ_WAIT_ANY', 0) / wc:5.1f} {100 * v.get('SQ_WAIT_INST_ANY', 0) / wc:6.1f} "
              f"{v.get('SQ_INSTS_VALU', 0) / max(waves, 1):9.0f} {v.get('SQ_INSTS_LDS', 0) / max(waves, 1):8.0f}")


if __name__ == "__main__":
    main()

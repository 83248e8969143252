#!/usr/bin/env python
"""Summarise a rocprofv3 --pmc CSV (pmc_counter_collection.csv): per kernel, mean of each counter.
usage: python tools/pmc_summary.py <dir or csv> [kernel substring]"""
import csv
import glob
import os
import sys
from collections import defaultdict

path = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
acc = defaultdict(lambda: defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if sub in k:
            acc[k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    wc = sum(cs.get("SQ_WAVE_CYCLES", [0])) or 1.0
    for c, v in sorted(cs.items()):
        s = sum(v)
        print(f"   {c:32s} n={len(v):4d} mean={s / len(v):16.1f}  /WAVE_CYCLES={s / wc:8.4f}")

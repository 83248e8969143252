#!/usr/bin/env python
"""Aggregate one rocprofv3 --pmc pass of SQ counters per kernel: share of wave cycles spent issuing (ACTIVE), parked in
s_waitcnt / barriers (WAIT_ANY), issue-stalled (WAIT_INST_ANY), and VALU instructions per wave.
A second directory (a pass with SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE) adds, per kernel: the share of the chip's
1024 matrix pipes that was executing an MFMA while the dispatch was active (MFMA busy cycles are summed over the SIMDs;
GRBM_GUI_ACTIVE counts the dispatch's chip-busy cycles) and the clock the chip held = GRBM_GUI_ACTIVE / the dispatch's
wall time (the kernel trace of the same pass).
usage: python tools/pmc_sq.py <dir with *counter_collection.csv> [<dir of the MFMA pass>]"""
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


N_SIMD = 256 * 4


def mfma_pass(d):
    """-> {kernel: (busy fraction of the matrix pipes, GHz)} from the SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE pass"""
    busy, gui, wall = collections.Counter(), collections.Counter(), collections.Counter()
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
                busy[k] += float(r["Counter_Value"])
            elif r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                gui[k] += float(r["Counter_Value"])
                if r.get("Start_Timestamp") and r.get("End_Timestamp"):
                    wall[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    if not wall:                      # older csv layouts keep the timestamps in the kernel trace only
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                wall[short(r["Kernel_Name"])] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    # GRBM_GUI_ACTIVE arrives summed over the 8 XCDs (checked on gfx950 / ROCm 7.2: conv_stream_kernel 2,609,971 counts for a
    # 152.5 us dispatch = 8 x 2.14 GHz); SQ_VALU_MFMA_BUSY_CYCLES is summed over the SIMDs and is exactly 32 per
    # v_mfma_f32_32x32x16_bf16 (148,635,648 for the 4,644,864 MFMAs of that dispatch)
    N_XCD = 8
    return {k: (busy[k] / (gui[k] / N_XCD * N_SIMD) if gui[k] else 0.0, gui[k] / N_XCD / wall[k] if wall.get(k) else 0.0) for k in gui}


def main():
    mf = mfma_pass(sys.argv[2]) if len(sys.argv) > 2 else {}
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
    print(f"{'kernel':58s} {'calls':>5s} {'waveMcyc':>9s} {'act%':>5s} {'wait%':>5s} {'stall%':>6s} {'VALU/wave':>9s} {'LDS/wave':>8s} {'MFMAbusy%':>9s} {'GHz':>5s}")
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
        wc = v.get("SQ_WAVE_CYCLES", 0.0)
        if wc <= 0:
            continue
        waves = v["_threads"] / 64.0
        print(f"{k[:58]:58s} {calls[k]:5d} {wc * 4 / 1e6:9.1f} {100 * v.get('SQ_ACTIVE_INST_ANY', 0) / wc:5.1f} "
              f"{100 * v.get('SQ_WAIT_ANY', 0) / wc:5.1f} {100 * v.get('SQ_WAIT_INST_ANY', 0) / wc:6.1f} "
              f"{v.get('SQ_INSTS_VALU', 0) / max(waves, 1):9.0f} {v.get('SQ_INSTS_LDS', 0) / max(waves, 1):8.0f} "
              f"{100 * mf.get(k, (0, 0))[0]:9.1f} {mf.get(k, (0, 0))[1]:5.2f}")
    if mf:
        print("MFMAbusy% = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs): share of the matrix pipes' cycles with an MFMA "
              "executing, at the clock the chip held; GHz = GRBM_GUI_ACTIVE / 8 / dispatch wall time (2.4 = the peak clock of the 2.5 PF figure)")


if __name__ == "__main__":
    main()

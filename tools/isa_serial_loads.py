"""Scans hipcc --save-temps assembly for SERIALISED memory loads: runs of (vector load, s_waitcnt vmcnt(0)) pairs, the
shape a per-element bounds test or a run-time branch around a load compiles to -- every load then pays a full memory
latency (round 3: phase_weight_kernel had 36 of them per workgroup, 229 us for a 21 MB weight).
    python tools/isa_serial_loads.py /tmp/st/*gfx950.s"""
import re
import sys

for f in sys.argv[1:]:
    s = open(f).read()
    for m in re.finditer(r"^(_Z\S+):[^\n]*\n(.*?)\.end_amdhsa_kernel", s, re.S | re.M):
        name, body = m.group(1), m.group(2).splitlines()
        ops = []
        for l in body:
            t = l.strip().split()
            if not t or t[0].startswith((";", ".")) and not t[0].startswith(".LBB"):
                continue
            ops.append(" ".join(t[:3]))
        best = run = 0
        i = 0
        last_load = -99
        for i, o in enumerate(ops):
            if re.match(r"(global_load|buffer_load|flat_load)", o):
                last_load = i
            elif o.startswith("s_waitcnt") and "vmcnt(0)" in o and i - last_load <= 3 and last_load >= 0:
                run += 1
                best = max(best, run)
                last_load = -99
            elif re.match(r"(v_mfma|s_barrier|global_store|buffer_store)", o):
                run = 0
        if best >= 3:
            short = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", name)[:90]
            print(f"{best:4d} serial (load, vmcnt(0)) pairs  {short}")

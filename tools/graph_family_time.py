#!/usr/bin/env python
"""Per-family kernel time INSIDE the replayed hipGraph of the default (overlapped) schedule, from one rocprofv3 kernel trace
(rocpd SQLite) of `bench.py`: the numbers that sit next to bench.py's serial-eager `roofline` figures.
Kernels of the two pullbacks share the CUs in this schedule, so a family's summed kernel time can exceed its share of the
wall clock; `wall_ms` is the union of all kernel intervals per step.
usage: python tools/graph_family_time.py <results.db> <steps_in_trace> <out.json>"""
import json
import re
import sqlite3
import sys

FAMILIES = [("conv3x3_fwd_dgrad", r"^(conv_stream_kernel|conv_phase4_kernel|conv_phase_kernel|conv_stream_mx8)"),
            ("conv_splitk_finish", r"^conv_splitk_finish"),
            ("conv_pointwise_1x1", r"^(conv_pw_kernel|conv_pw2_kernel|stem_conv|stem_dgrad)"),
            ("conv_other_fwd_dgrad", r"^(conv_patch|conv_igemm|expand_taps)"),
            ("wgrad", r"^(conv_wgrad|wgrad_reduce)"),
            ("norm", r"^(bn_|cbn_|reduce_rows)"),
            ("losses_attention", r"^(wl_|cl_|attn_|xent|hinge|proj_|word_)"),
            ("optimiser_spectral", r"^(adam_|sn_|wprep_)"),
            ("torch", r"^(at::|__amd_rocclr)"),
            ("rccl", r"^(ncclDevKernel|rccl)")]


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return re.sub(r"^void ", "", name)


def main():
    db, steps, out = sys.argv[1], float(sys.argv[2]), sys.argv[3]
    rows = sqlite3.connect(db).execute("select name, start, end from kernels").fetchall()
    fam = {}
    for name, s, e in rows:
        n = short(name)
        key = next((k for k, pat in FAMILIES if re.match(pat, n)), "other")
        d = fam.setdefault(key, dict(launches=0, ms=0.0))
        d["launches"] += 1
        d["ms"] += (e - s) / 1e6
    iv = sorted((s, e) for _, s, e in rows)
    wall, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                wall += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        wall += cur_e - cur_s
    res = {"schedule": "default (overlapped pullbacks, prefetched generator forward), replayed hipGraph", "steps_in_trace": steps,
           "families": {k: dict(launches_per_step=round(v["launches"] / steps, 1), ms_per_step=round(v["ms"] / steps, 3)) for k, v in
                        sorted(fam.items(), key=lambda kv: -kv[1]["ms"])},
           "sum_kernel_ms_per_step": round(sum(v["ms"] for v in fam.values()) / steps, 3),
           "gpu_busy_wall_ms_per_step": round(wall / 1e6 / steps, 3)}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()

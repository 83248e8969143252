#!/usr/bin/env python
"""MODELLED multi-GPU throughput of the C1 step ("report 1-GPU measured, and for 2/4/8 ... modelled =
compute(1 GPU) + max(0, comm - overlappable) ... clearly labelled as modelled").  Nothing here is a measurement of more
than one GPU: the 1-GPU inputs are measured (profiles/<tag>_*), the link bandwidth is an ASSUMPTION stated per row.

  compute     the data-parallel schedule's step time at world size 1 under torchrun with the RCCL calls inside the graph
              (profiles/<tag>_bench_c1_torchrun_world1_graph_dp_overlap.json): it already contains the schedule's own cost
              (no prefetched generator forward, deferred discriminator update);
  comm        ring / direct all-reduce of S bytes over N GPUs moves 2 (N - 1) / N * S bytes per GPU at `bw` per GPU + 30 us
              per bucket launch; volumes: D 352 MB twice, G 314 MB once (float32), halved with --grad-transport bf16;
  windows     round 4: BOTH networks' exchanges are issued in slices from inside their backward passes (the gradient through
              sigma moved behind the exchange, into the optimiser kernel) -- D: [DiscBlock_4 + heads 172 MB] after the first
              block of the backward, [DiscBlock_3 132 MB] after the second, [rest 48 MB] at the end; G: three slices along its
              backward.  train_d: the slices have the rest of D's backward (~6 ms of the trace) plus, deferred, train_g_d's
              generator forward (4.0 ms); train_g_d: D's and G's slices share the links during the two pullbacks (~8.5 ms from
              the first ready slice to the end), the last G slice (105 MB) is exposed;
  contention  the collective's kernels share CUs and HBM with the step: measured with a stand-in copy kernel on k workgroups
              beside the replayed step (tools/cu_contention.py -> profiles/<tag>_cu_contention.txt); the slowdown at the copy
              rate nearest the link rate (x2: an all-reduce reads and writes) is charged for the fraction of the step during
              which exchanges are in flight.
  exclusive   (round 6) dp.GradSync(schedule="exclusive"): each arena exchanged in one piece after its half step's backward passes,
              nothing beside the RCCL kernels -- step = the exclusive schedule's world-1 time (profiles/<tag>_bench_c1_torchrun_
              world1_graph_dp_exclusive.json) + ALL of comm, no contention term.  The schedule that cannot lose to the overlap going
              wrong; bench.py --grad-schedule auto (the default for N > 1) times both on the node and keeps the faster.
usage: python tools/model_scaling.py [tag]   (default r04)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def last_json(path):
    return json.loads([l for l in open(path).read().strip().splitlines() if l.startswith("{")][-1])


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
    prof = os.path.join(ROOT, "profiles")
    one = last_json(os.path.join(prof, f"{tag}_bench_c1.json"))
    dp1 = last_json(os.path.join(prof, f"{tag}_bench_c1_torchrun_world1_graph_dp_overlap.json"))
    t1, tdp = one["ms_per_step"], dp1["ms_per_step"]
    epath = os.path.join(prof, f"{tag}_bench_c1_torchrun_world1_graph_dp_exclusive.json")
    texc = last_json(epath)["ms_per_step"] if os.path.exists(epath) else None
    cont = None
    cpath = os.path.join(prof, f"{tag}_cu_contention.txt")
    if os.path.exists(cpath):
        c = last_json(cpath)
        base = c["ms_per_step"]["0"]
        cont = sorted((int(k) * c["copy_GBps_per_workgroup"], v / base - 1.0) for k, v in c["ms_per_step"].items())
    print(f"MODELLED -- not measured on more than one GPU.  inputs ({tag}): 1-GPU step {t1:.2f} ms ({one['value']:.0f} img/s), "
          f"data-parallel schedule at world 1 {tdp:.2f} ms" + ("" if cont is None else
          "; contention (copy GB/s -> step slowdown): " + ", ".join(f"{r:.0f} -> +{100 * p:.1f} %" for r, p in cont)))

    def penalty(rate):
        if cont is None:
            return 0.0
        for (r0, p0), (r1, p1) in zip(cont, cont[1:]):
            if r0 <= rate <= r1:
                return p0 + (p1 - p0) * (rate - r0) / max(r1 - r0, 1e-9)
        return cont[-1][1] if rate > cont[-1][0] else 0.0
    halves = {"train_d": dict(bytes=[172e6, 132e6, 48e6], window=10.0, tail=0.0),
              "train_g_d": dict(bytes=[172e6, 132e6, 48e6, 105e6, 104e6], window=8.5, tail=105e6)}
    if texc is not None:
        print(f"exclusive schedule at world 1 {texc:.2f} ms")
    print(f"{'schedule':10s} {'transport':9s} {'GB/s per GPU (assumed)':>26s} | " + " | ".join(f"N={n}:   ms   img/s  eff" for n in (2, 4, 8)))
    for transport, scale in (("float32", 1.0), ("bf16", 0.5)):
        for bw, what in ((153e9, "153 (one xGMI link, ring)"), (300e9, "300 (all links, direct)")):
            if texc is not None:
                cells = []
                for n in (2, 4, 8):
                    f = 2.0 * (n - 1) / n * scale / bw * 1e3
                    comm = sum(b * f + 0.03 for b in (352e6, 352e6, 314e6))      # D twice, G once, one piece each: all of it exposed
                    step = texc + comm + (0.6 if transport == "bf16" else 0.0)
                    ips = 56 * n / step * 1e3
                    cells.append(f"{step:6.2f} {ips:7.0f} {ips / (n * one['value']):5.2f}")
                print(f"{'exclusive':10s} {transport:9s} {what:>26s} | " + " | ".join(cells))
            cells = []
            for n in (2, 4, 8):
                f = 2.0 * (n - 1) / n * scale / bw * 1e3
                exposed = busy = 0.0
                for h in halves.values():
                    link = sum(b * f + 0.03 for b in h["bytes"])
                    tail = h["tail"] * f
                    exposed += max(0.0, link - h["window"]) + tail
                    busy += link + tail
                extra = 0.6 if transport == "bf16" else 0.0          # two cast passes over ~1 GB of gradients
                slow = penalty(2.0 * bw / 1e9) * min(1.0, busy / tdp)   # the copy kernels run only while exchanges are in flight
                step = tdp * (1.0 + slow) + exposed + extra
                ips = 56 * n / step * 1e3
                cells.append(f"{step:6.2f} {ips:7.0f} {ips / (n * one['value']):5.2f}")
            print(f"{'overlapped':10s} {transport:9s} {what:>26s} | " + " | ".join(cells))
    print("target (BASELINE.json): >= 6.5x at 8 GPUs = efficiency 0.81")


if __name__ == "__main__":
    main()

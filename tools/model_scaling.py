#!/usr/bin/env python
"""MODELLED multi-GPU throughput of the C1 step ("report 1-GPU measured, and for 2/4/8 ... modelled =
compute(1 GPU) + max(0, comm - overlappable) ... clearly labelled as modelled").  Nothing here is a measurement of more
than one GPU: the 1-GPU inputs are measured (profiles/r03_*), the link bandwidth is an ASSUMPTION stated per column.

  compute    the data-parallel schedule's step time at world size 1 under torchrun with the RCCL calls inside the graph
             (profiles/r03_bench_c1_torchrun_world1_graph_dp_overlap.json) -- it already contains the schedule's own cost
             (no prefetched generator forward, deferred discriminator update);
  comm       ring / direct all-reduce of S bytes over N GPUs moves 2 (N - 1) / N * S bytes per GPU at `bw` per GPU
             + 30 us per bucket launch; volumes: D 352 MB twice, G 314 MB once (float32), halved with --grad-transport bf16;
  windows    what each exchange can hide behind (serial kernel times of the r03 trace, scaled by the measured overlap of
             the two-stream schedule): train_d's D exchange under train_g_d's generator forward (4.0 ms); train_g_d's D
             exchange under the rest of the g-stream once the d-stream has finished (2.5 ms in the overlapped-pullback
             schedule); G's first two buckets under the rest of the generator backward (3.0 ms), the third bucket exposed.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    prof = os.path.join(ROOT, "profiles")
    one = json.loads(open(os.path.join(prof, "r03_bench_c1.json")).read().strip().splitlines()[-1])
    dp1 = json.loads(open(os.path.join(prof, "r03_bench_c1_torchrun_world1_graph_dp_overlap.json")).read().strip().splitlines()[-1])
    t1, tdp = one["ms_per_step"], dp1["ms_per_step"]
    print(f"MODELLED -- not measured on more than one GPU.  inputs: 1-GPU step {t1:.2f} ms ({one['value']:.0f} img/s), "
          f"data-parallel schedule at world 1 {tdp:.2f} ms")
    vols = {"d1": 352e6, "d2": 352e6, "g12": 314e6 * 2 / 3, "g3": 314e6 / 3}
    win = {"d1": 4.0, "d2": 2.5, "g12": 3.0, "g3": 0.0}
    print(f"{'transport':9s} {'GB/s per GPU (assumed)':>24s} | " + " | ".join(f"N={n}: ms  img/s  eff" for n in (2, 4, 8)))
    for transport, scale in (("float32", 1.0), ("bf16", 0.5)):
        for bw, what in ((153e9, "153 (one xGMI link, ring)"), (300e9, "300 (all links, direct)")):
            cells = []
            for n in (2, 4, 8):
                exposed = 0.0
                for k, s in vols.items():
                    t = 2.0 * (n - 1) / n * s * scale / bw * 1e3 + 0.03 * max(1, round(s / 128e6))
                    exposed += max(0.0, t - win[k])
                extra = 0.6 if transport == "bf16" else 0.0          # two cast passes over ~1 GB of gradients
                step = tdp + exposed + extra
                ips = 56 * n / step * 1e3
                cells.append(f"{step:6.2f} {ips:7.0f} {ips / (n * one['value']):5.2f}")
            print(f"{transport:9s} {what:>24s} | " + " | ".join(cells))
    print("target (BASELINE.json): >= 6.5x at 8 GPUs = efficiency 0.81")


if __name__ == "__main__":
    main()

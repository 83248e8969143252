#!/usr/bin/env python
"""Which torch-launched kernels (at::*, rocclr copies / fills) does the replayed step still contain?  From a rocprofv3 kernel trace:
grouped by (kernel, workgroups), calls per step and time.   usage: python tools/torch_kernels_in_trace.py <results.db> <steps>"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    g = "grid_x" if "grid_x" in cols else "grid_size_x"
    w = "workgroup_x" if "workgroup_x" in cols else "workgroup_size_x"
    agg = {}
    for name, d, wgs in cur.execute(f"select name, end - start, {g} / {w} from kernels"):
        if not ("at::" in name or "rocclr" in name or "Cijk" in name):
            continue
        m = re.search(r"(elementwise_kernel_manual_unroll|vectorized_elementwise_kernel|CatArrayBatchedCopy\w*|__amd_rocclr_\w+|reduce_kernel|index\w*|Cijk\w{0,20})", name)
        fn = re.search(r"(direct_copy_kernel_cuda|bfloat16_copy_kernel_cuda|FillFunctor<[^>]+>|CUDAFunctor_add|MulFunctor|\w+Functor\w*|\w+_kernel_cuda)", name)
        key = ((m.group(1) if m else name[:40]) + " " + (fn.group(1) if fn else ""), int(wgs))
        a = agg.setdefault(key, [0, 0])
        a[0] += 1
        a[1] += d
    tot = sum(a[1] for a in agg.values())
    print(f"torch-launched kernels: {sum(a[0] for a in agg.values()) / steps:.1f} launches, {tot / 1e6 / steps:.3f} ms per step")
    print(f"{'kernel':78s} {'wgs':>7s} {'/step':>6s} {'avg_us':>7s} {'ms/step':>8s}")
    for (k, wgs), (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{k[:78]:78s} {wgs:7d} {n / steps:6.1f} {d / n / 1e3:7.1f} {d / 1e6 / steps:8.3f}")


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""One-line step timing for A/B runs: ms/step of the default workload and of the G/D step alone (graph replay)."""
import json, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--steps", "20"] + sys.argv[1:],
                     capture_output=True, text=True).stdout.strip().splitlines()[-1]
d = json.loads(out)
r = d.get("roofline") or {}
print(f"step {d['ms_per_step']:.2f} ms | gd_only {d.get('gd_only', {}).get('ms_per_step', float('nan')):.2f} ms | conv_stream "
      f"{r.get('ms_per_step', 0):.2f} ms {r.get('achieved', 0):.0f} TF/s | wgrad {(r.get('wgrad') or {}).get('ms_per_step', 0):.2f} ms "
      f"{(r.get('wgrad') or {}).get('achieved', 0):.0f} TF/s | family {(r.get('family') or {}).get('ms_per_step', 0):.2f} ms")

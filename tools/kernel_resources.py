#!/usr/bin/env python
"""Per-kernel register / scratch usage of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel.
usage: python tools/kernel_resources.py xmcgan_image_generation_amd/csrc/conv_stream.hip [extra hipcc flags]"""
import re
import subprocess
import sys

src, extra = sys.argv[1], sys.argv[2:]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-result", "-c", src, "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage"] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:") or t.startswith("Name:"):
        cur = t.split(":", 1)[1].strip()
        rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1)
        rows[cur][k.strip()] = v.strip()
dem = subprocess.run(["c++filt"], input="\n".join(rows), capture_output=True, text=True).stdout.splitlines()
for name, nice in zip(rows, dem):
    r = rows[name]
    nice = re.sub(r"\(anonymous namespace\)::", "", nice)
    nice = re.sub(r"\(.*", "", nice).replace("void ", "")
    print(f"{nice[:60]:60s} VGPR {r.get('VGPRs', '?'):>4s} AGPR {r.get('AGPRs', '?'):>4s} spillV {r.get('VGPRs Spill', '?'):>3s} spillS {r.get('SGPRs Spill', '?'):>3s} "
          f"scratch {r.get('ScratchSize [bytes/lane]', '?'):>4s} LDS {r.get('LDS Size [bytes/block]', '?'):>6s} occ {r.get('Occupancy [waves/SIMD]', '?')}")

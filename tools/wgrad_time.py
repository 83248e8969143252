"""One line of weight-gradient timings (3x3, bf16) for library A/B and ablation builds (-DWG_ABL=n, conv_wgrad_dma.hip).
usage (GPU box): PYTHONPATH=. python tools/wgrad_time.py [zeros]"""
import sys
import torch
from xmcgan_image_generation_amd.ops import HipOps

ops = HipOps(torch.bfloat16)
g = torch.Generator().manual_seed(0)
zeros = len(sys.argv) > 1 and sys.argv[1] == "zeros"
out = []
for (n, h, cin, cout) in ((56, 128, 96, 96), (56, 64, 192, 192), (56, 32, 384, 384), (56, 16, 768, 768), (56, 8, 1536, 1536)):
    x = torch.randn((n, h, h, cin), generator=g).cuda().bfloat16()
    dy = torch.randn((n, h, h, cout), generator=g).cuda().bfloat16()
    if zeros:
        x.zero_(); dy.zero_()
    dw = torch.zeros((cout, 9, cin), device="cuda")
    fl = 2 * n * h * h * cin * cout * 9
    best = 1e9
    for r in range(4):
        for _ in range(3):
            ops.conv_wgrad(x, dy, dw, ks=3, sync=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.conv_wgrad(x, dy, dw, ks=3, sync=True)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20)
    out.append(f"{h}^2 {cin}: {best * 1e3:5.0f} us {fl / best / 1e9:5.0f}")
print(" | ".join(out))

import torch, math
from xmcgan_image_generation_amd.ops import HipOps
ops = HipOps(torch.bfloat16)
for c in (96, 384, 1536):
    w = torch.randn((c, 9, c), device="cuda")
    for _ in range(3): ops.prep_conv_weight(w, None, True, phase="ups")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.prep_conv_weight(w, None, True, phase="ups")
    e1.record(); torch.cuda.synchronize()
    ops.phase_conv = False
    e2, e3 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e2.record()
    for _ in range(20): ops.prep_conv_weight(w, None, True, phase="ups")
    e3.record(); torch.cuda.synchronize()
    ops.phase_conv = True
    print(c, "prep+phase", e0.elapsed_time(e1) / 20 * 1e3, "us; prep only", e2.elapsed_time(e3) / 20 * 1e3, "us")

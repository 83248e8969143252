import torch
from xmcgan_image_generation_amd.ops import HipOps
ops = HipOps(torch.bfloat16)
g = torch.Generator().manual_seed(0)
def timed(fn):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 50 * 1e3
for (m, k, n) in ((56, 256, 3072), (56, 256, 1536), (56, 768, 256), (56, 896, 24576), (112, 1536, 1), (112, 768, 1536)):
    a = torch.randn((m, k), generator=g).cuda(); b = torch.randn((k, n), generator=g).cuda(); dy = torch.randn((m, n), generator=g).cuda()
    out = torch.zeros((m, n), device="cuda"); dw = torch.zeros((k, n), device="cuda")
    r = []
    for fast in (False, True):
        r.append(timed(lambda: ops.gemm(a, b, beta=1.0, out=out, fast=fast)))
        r.append(timed(lambda: ops.gemm(a, dy, ta=True, beta=1.0, out=dw, fast=fast)))
        r.append(timed(lambda: ops.gemm(dy, b, tb=True, fast=fast)))
    print(f"{m}x{k}x{n}: f32 fwd {r[0]:.0f} wgrad {r[1]:.0f} dgrad {r[2]:.0f} | bf16mfma fwd {r[3]:.0f} wgrad {r[4]:.0f} dgrad {r[5]:.0f} us")

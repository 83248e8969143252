import torch, math
from xmcgan_image_generation_amd.ops import HipOps
ops = HipOps(torch.bfloat16)
g = torch.Generator().manual_seed(0)
def timed(fn):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20
for (n, h, cin, cout, kw) in ((112, 128, 96, 96, dict(dy_ups=True, alpha=0.25)), (112, 64, 96, 192, {}), (56, 64, 192, 192, {}), (56, 32, 384, 192, dict(x_ups=True)), (112, 16, 768, 768, dict(dy_ups=True, alpha=0.25)), (56, 128, 96, 96, {})):
    x = torch.randn((n, h, h, cin), generator=g).cuda().bfloat16()
    ho = 2 * h if kw.get("x_ups") else h
    hd = ho // 2 if kw.get("dy_ups") else ho
    dy = torch.randn((n, hd, hd, cout), generator=g).cuda().bfloat16()
    dw = torch.zeros((cout, 9, cin), device="cuda"); db = torch.zeros((cout,), device="cuda")
    a = timed(lambda: ops.conv_wgrad(x, dy, dw, db, ks=3, sync=True, **kw))
    b = timed(lambda: ops.conv_wgrad(x, dy, dw, None, ks=3, sync=True, **kw))
    print(f"{n}x{h}^2 {cin}>{cout} {kw}: wgrad with db {a*1e3:.0f} us / without {b*1e3:.0f} us")

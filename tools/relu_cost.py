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
for (n, h, cin, cout, dy_ups) in ((112, 128, 96, 96, True), (112, 64, 192, 192, True), (112, 32, 384, 384, True), (112, 64, 96, 192, False), (112, 32, 192, 384, False), (112, 16, 768, 768, True)):
    x = torch.randn((n, h, h, cin), generator=g).cuda().bfloat16()
    hd = h // 2 if dy_ups else h
    dy = torch.randn((n, hd, hd, cout), generator=g).cuda().bfloat16()
    dw = torch.zeros((cout, 9, cin), device="cuda"); db = torch.zeros((cout,), device="cuda")
    a = timed(lambda: ops.conv_wgrad(x, dy, dw, db, ks=3, x_relu=True, dy_ups=dy_ups, alpha=0.25, sync=True))
    b = timed(lambda: ops.conv_wgrad(x, dy, dw, db, ks=3, x_relu=False, dy_ups=dy_ups, alpha=0.25, sync=True))
    w = torch.randn((cout, 9, cin), generator=g) / math.sqrt(9 * cin)
    wf, _ = ops.prep_conv_weight(w.cuda(), None, True, phase="pool" if dy_ups else None)
    if dy_ups:
        c = timed(lambda: ops.conv(x, wf, None, ks=3, pool_out=True, relu_in=True)); d = timed(lambda: ops.conv(x, wf, None, ks=3, pool_out=True))
    else:
        c = timed(lambda: ops.conv(x, wf, None, ks=3, relu_in=True)); d = timed(lambda: ops.conv(x, wf, None, ks=3))
    print(f"{n}x{h}^2 {cin}>{cout} dy_ups={dy_ups}: wgrad x_relu {a*1e3:.0f} / plain {b*1e3:.0f} us | fwd relu_in {c*1e3:.0f} / plain {d*1e3:.0f} us")

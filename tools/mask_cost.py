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
# D 128>64 c1.dgrad (phase out): dP (112,64,64,96) -> (112,128,128,96), mask h1
w = torch.randn((96, 9, 96), generator=g) / 30
wf, wd = ops.prep_conv_weight(w.cuda(), None, True, phase="pool")
dp = torch.randn((112, 64, 64, 96), generator=g).cuda().bfloat16()
h1 = torch.randn((112, 128, 128, 96), generator=g).cuda().bfloat16()
for r in range(3):
    a = timed(lambda: ops.conv(dp, wd, None, ks=3, ups=True, alpha=0.25, mask=h1))
    b = timed(lambda: ops.conv(dp, wd, None, ks=3, ups=True, alpha=0.25))
    print(f"D128 c1.dgrad  mask {a*1e3:.0f} us   no mask {b*1e3:.0f} us")
# D 64: (112,32,32,192)->(112,64,64,192)
w = torch.randn((192, 9, 192), generator=g) / 40
wf, wd = ops.prep_conv_weight(w.cuda(), None, True, phase="pool")
dp = torch.randn((112, 32, 32, 192), generator=g).cuda().bfloat16()
h1 = torch.randn((112, 64, 64, 192), generator=g).cuda().bfloat16()
a = timed(lambda: ops.conv(dp, wd, None, ks=3, ups=True, alpha=0.25, mask=h1)); b = timed(lambda: ops.conv(dp, wd, None, ks=3, ups=True, alpha=0.25))
print(f"D64 c1.dgrad  mask {a*1e3:.0f} us   no mask {b*1e3:.0f} us")
# ResNet bwd 1x1: (56,64,64,256)->64 mask ; (56,64,64,64)->256 mask res
for (cin, cout, res) in ((256, 64, False), (64, 256, True), (512, 128, False), (128, 512, True)):
    hh = 64 if cin in (256, 64) else 32
    w = torch.randn((cout, 1, cin), generator=g) / math.sqrt(cin)
    wf, wd = ops.prep_conv_weight(w.cuda(), None, True)
    x = torch.randn((56, hh, hh, cin), generator=g).cuda().bfloat16()
    m = torch.randn((56, hh, hh, cout), generator=g).cuda().bfloat16()
    rr = torch.randn((56, hh, hh, cout), generator=g).cuda().bfloat16() if res else None
    a = timed(lambda: ops.conv(x, wf, None, ks=1, mask=m, res=rr, mask_after_res=res)); b = timed(lambda: ops.conv(x, wf, None, ks=1, res=rr))
    print(f"pw {hh}^2 {cin}>{cout} res={res}: mask {a*1e3:.0f} us   no mask {b*1e3:.0f} us")

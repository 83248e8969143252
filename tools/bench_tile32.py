import os, sys, torch
sys.path.insert(0, "/root/repo")
from xmcgan_image_generation_amd.ops import HipOps
g = torch.Generator().manual_seed(0)
def run(tile32):
    ops = HipOps(dtype=torch.bfloat16); ops.tile32 = tile32
    x = torch.randn((56,128,128,96), generator=g).to(torch.bfloat16).cuda()
    w = (torch.randn((3,9,96), generator=g)/30).cuda()
    wf,_ = ops.prep_conv_weight(w)
    b = torch.randn((3,), generator=g).cuda()
    res = torch.randn((56,64,64,3), generator=g).to(torch.bfloat16).cuda()
    outs=[]
    for kw in (dict(bias=b), dict(bias=None, res=res, res_ups=True, res_scale=0.25)):
        bias = kw.pop("bias")
        f = lambda: ops.conv(x, wf, bias, ks=3, **kw)
        y = f(); torch.cuda.synchronize()
        s,e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(50): f()
        e.record(); torch.cuda.synchronize()
        outs.append((y.clone(), s.elapsed_time(e)/50*1e3))
    return outs, x, w, b, res
g = torch.Generator().manual_seed(0); a,x,w,b,res = run(False)
g = torch.Generator().manual_seed(0); c,_,_,_,_ = run(True)
for (ya,ta),(yc,tc) in zip(a,c):
    print(f"128-cout tile {ta:7.1f} us   32-cout tile {tc:7.1f} us   equal bytes: {torch.equal(ya,yc)}  max diff {float((ya.float()-yc.float()).abs().max()):.3g}")
ref = torch.nn.functional.conv2d(x.float().permute(0,3,1,2), w.view(3,3,3,96).permute(0,3,1,2).to(torch.bfloat16).float(), b, padding=1).permute(0,2,3,1)
print("vs F.conv2d:", float((c[0][0].float()-ref).abs().max()), float(ref.abs().max()))

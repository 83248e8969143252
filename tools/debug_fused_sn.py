"""debug: per-entry comparison of the fused power iteration against float64 torch (GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_fused_opt import _small_d

cfg, gen, disc, state = _small_d()
d = disc(train=True)
ops = d.ops
params, sn = state.d_optimizer.target, state.discriminator_state["spectral_norm_stats"]
arena = d._bind(params)
u0 = d._pack_u0(sn)
bufs, part = ops.wprep_run(d.wp, arena.params, u0)
torch.cuda.synchronize()
print("wp n", d.wp["n"], "blocks", d.wp["blocks"], "blocks_c", d.wp["blocks_c"], "part", d.wp["part"], "irr n", d.irr["n"] if d.irr else 0)
off = 0
for k, e in enumerate(d.wp["entries"]):
    cout, cin, taps = e["cout"], e["cin"], e["taps"]
    cols = taps * cin
    w = arena.params[e["w_off"]:e["w_off"] + cout * cols].view(cout, cols).double()
    ref = u0[e["u_off"]:e["u_off"] + cout].double() @ w
    got = part[off:off + (cout // 32) * cols].view(cout // 32, cols).double().sum(0)
    off += (cout // 32) * cols
    print(f"wprep entry {k} site {e['site']} cout {cout} cin {cin} taps {taps} u_off {e['u_off']} v_off {e['v_off']} partial err "
          f"{float((got - ref).abs().max()) / float(ref.abs().max()):.2e}")
u_new, v, scal = ops.sn_bank_power_iter_fused(d.bank, d.irr, d.wp, arena.params, u0, part)
ru, rv, rs = ops.sn_bank_power_iter(d.bank, arena.params, u0)
torch.cuda.synchronize()
for i, (s, e) in enumerate(zip(d.sn_sites, d.bank["entries"])):
    rows, cols = e["rows"], e["cols"]
    w = arena.params[e["w_off"]:e["w_off"] + rows * cols].view(rows, cols).double()
    uu = u0[e["u_off"]:e["u_off"] + e["nu"]].double()
    vr = (uu @ w) if e["u_axis"] == 0 else (w @ uu)
    vn = vr * torch.rsqrt((vr * vr).sum() + 1e-10)
    ur = (w @ vn) if e["u_axis"] == 0 else (vn @ w)
    un = ur * torch.rsqrt((ur * ur).sum() + 1e-10)
    sg = float((ur * un).sum())
    fv, lv = v[e["v_off"]:e["v_off"] + e["nv"]].double(), rv[e["v_off"]:e["v_off"] + e["nv"]].double()
    fu, lu = u_new[e["u_off"]:e["u_off"] + e["nu"]].double(), ru[e["u_off"]:e["u_off"] + e["nu"]].double()
    print(f"entry {i:2d} {s.path:34s} rows {rows:5d} cols {cols:6d} axis {e['u_axis']} | v err fused {float((fv - vn).abs().max()):.2e} legacy {float((lv - vn).abs().max()):.2e}"
          f" | u err fused {float((fu - un).abs().max()):.2e} legacy {float((lu - un).abs().max()):.2e} | sigma ref {sg:.5f} fused {float(scal[2 * i]):.5f} legacy {float(rs[2 * i]):.5f}")

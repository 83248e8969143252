"""Is v_pk_add_f32 with crossed halves (op_sel:[0,1] op_sel_hi:[1,0]) itself unreliable on gfx950 when other kernels share the CUs?
(the form the MX-fp8 kernel's residual add had been compiled to.)  xmc_pk_add_cross_probe runs the exact instruction pair
in a pure-VALU kernel against scalar arithmetic, alone and beside a weight-gradient / convolution neighbour on a second stream.
    PYTHONPATH=. python tools/pk_add_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    from xmcgan_image_generation_amd import _lib
    from xmcgan_image_generation_amd.ops import HipOps, _p
    PROBE = _lib.load_probe()
    ops = HipOps(dtype=torch.bfloat16)
    g = torch.Generator().manual_seed(0)
    side = torch.cuda.Stream()
    nx = torch.randn((32, 64, 64, 192), generator=g).to(torch.bfloat16).cuda()
    nw = (torch.randn((192, 9, 192), generator=g) / 42).cuda()
    nwf, _ = ops.prep_conv_weight(nw)
    ndw = torch.zeros((192, 9, 192), device="cuda")
    ndb = torch.zeros((192,), device="cuda")

    n1 = torch.randn((32, 64, 64, 192), generator=g).to(torch.bfloat16).cuda()
    w1 = (torch.randn((192, 1, 192), generator=g) / 14).cuda()
    w1f, _ = ops.prep_conv_weight(w1)
    dw1 = torch.zeros((192, 1, 192), device="cuda")
    src = torch.zeros((1 << 22,), dtype=torch.float32, device="cuda")
    outp = torch.zeros((4,), device="cuda")

    def neighbour(kind):
        if kind == "alone":
            return
        with torch.cuda.stream(side):
            for _ in range(10):
                if kind == "wgrad":
                    ops.conv_wgrad(nx, nx, ndw, ndb, ks=3, x_relu=True, sync=True)
                elif kind == "wgrad, no relu pass":
                    ops.conv_wgrad(nx, nx, ndw, None, ks=3, x_relu=False, sync=True)
                elif kind == "wgrad 1x1":
                    ops.conv_wgrad(n1, n1, dw1, None, ks=1, sync=True)
                elif kind == "pointwise conv (LDS-DMA)":
                    ops.conv(n1, w1f, None, ks=1)
                elif kind == "LDS-DMA load ring only":
                    PROBE.xmc_load_path_probe(1 | (3 << 4), 512, 400, _p(src), src.numel() * 4, _p(outp), ops._stream())
                elif kind == "register load ring only":
                    PROBE.xmc_load_path_probe(0 | (3 << 4), 512, 400, _p(src), src.numel() * 4, _p(outp), ops._stream())
                elif kind == "MFMA only":
                    PROBE.xmc_mfma_rate_probe(0, 512, 2000, _p(outp), ops._stream())
                else:
                    ops.conv(nx, nwf, None, ks=3)
    bad = torch.zeros((1,), dtype=torch.int32, device="cuda")
    if "--classes" in sys.argv:
        names = {1: "MFMA 32x32x16, dependent chain", 1024: "MFMA 32x32x16, 4 independent accumulators", 2048: "MFMA 16x16x32, dependent chain", 2: "v_dot2c_f32_bf16", 4: "ds_read_b64_tr_b16", 8: "ds_write/read_b128", 16: "v_pk_max_i16", 32: "LDS-DMA",
                 64: "s_barrier", 128: "v_permlane32_swap", 256: "v_cvt_pk_bf16_f32", 512: "global loads"}
        masks = list(names) + [1 | 32 | 64, 1 | 8 | 64, 1024 | 8 | 64, 2 | 4, 1023]
        for m in masks:
            tot = 0
            for _ in range(10):
                bad.zero_()
                torch.cuda.synchronize()
                with torch.cuda.stream(side):
                    for _ in range(4):
                        assert PROBE.xmc_class_neighbour(m, 1024, 3000, _p(src), src.numel() * 4, _p(outp), ops._stream()) == 0
                assert PROBE.xmc_pk_add_cross_probe(0, 2048, 4000, _p(bad), ops._stream()) == 0
                torch.cuda.synchronize()
                tot += int(bad.item())
            tag = " + ".join(v for k, v in names.items() if m & k) if m != 1023 else "all of them"
            print(f"crossed v_pk_add_f32 beside a loop of {tag:60s}: mismatching rounds in 10 launches: {tot}", flush=True)
        return
    for mode, tag in ((0, "crossed v_pk_add_f32"), (1, "uncrossed (operands swapped by hand)"),
                      (2, "v_pk_fma_f32 op_sel:[1,0,0]"), (3, "v_pk_fma_f32 op_sel_hi:[0,1,1], SGPR"),
                      (4, "v_pk_fma_f32 crossed src0")):
        for kind in (("alone", "conv", "wgrad", "wgrad, no relu pass", "wgrad 1x1", "pointwise conv (LDS-DMA)", "LDS-DMA load ring only",
                      "register load ring only", "MFMA only") if mode == 0 else ("alone", "wgrad", "pointwise conv (LDS-DMA)")):
            tot = 0
            for _ in range(20):
                bad.zero_()
                torch.cuda.synchronize()
                neighbour(kind)
                rc = PROBE.xmc_pk_add_cross_probe(mode, 2048, 4000, _p(bad), ops._stream())
                assert rc == 0
                torch.cuda.synchronize()
                tot += int(bad.item())
            print(f"{tag:40s} neighbour {kind:26s}: mismatching rounds in 20 launches x 2048 x 256 lanes x 4000 rounds: {tot}", flush=True)


if __name__ == "__main__":
    main()

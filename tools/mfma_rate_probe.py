"""Sustained matrix-core rate of this box: registers-only MFMA loops (xmc_mfma_rate_probe) at the occupancies the
convolution kernels run at.  The roofline in bench.py prices against the datasheet peak (2.5 PFLOP/s bf16 dense at
2.4 GHz); this prints what a kernel with NO memory instruction sustains, i.e. the clock the chip holds under MFMA load.
usage (GPU box): python tools/mfma_rate_probe.py"""
import torch
from xmcgan_image_generation_amd import _lib

lib = _lib.load_probe()
out = torch.zeros(16, device="cuda")
st = torch.cuda.current_stream().cuda_stream
FLOP = {0: 2 * 32 * 32 * 16, 1: 2 * 32 * 32 * 64}
OPS = {0: "random", 1: "constant"}
NAME = {0: "bf16 32x32x16", 1: "mx-fp8 32x32x64 (scaled)"}
for mode, const in ((0, 0), (0, 1), (1, 0), (1, 1)):
    for wg_per_cu in (1, 2):
        for iters in (2000, 20000):
            blocks = 256 * wg_per_cu
            _lib.check(lib.xmc_mfma_rate_probe(mode | (const << 1), blocks, 200, out.data_ptr(), st), "probe")
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                _lib.check(lib.xmc_mfma_rate_probe(mode | (const << 1), blocks, iters, out.data_ptr(), st), "probe")
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1))
            fl = blocks * 4 * 8 * iters * FLOP[mode]
            # cycles per MFMA per SIMD if the pipe never idles: 4 SIMDs / CU, wg_per_cu waves each
            print(f"{NAME[mode]:26s} {OPS[const]:8s} operands {wg_per_cu} wave(s)/SIMD iters {iters:6d}: {best:8.3f} ms  {fl / best / 1e9:7.0f} TFLOP/s"
                  f"  -> {fl / best / 1e9 / (256 * 4 * FLOP[mode] / (32 if mode == 0 else 64)) * 1e3:.2f} GHz-equivalent")

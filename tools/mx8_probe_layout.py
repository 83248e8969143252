#!/usr/bin/env python
"""Empirical operand / scale layout of v_mfma_scale_f32_32x32x64_f8f6f4 through xmc_mx8_probe (GPU box)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from xmcgan_image_generation_amd import _lib
lib = _lib.load()

def tab():
    t = np.zeros(256)
    for b in range(256):
        s, e, m = b >> 7, (b >> 3) & 0xF, b & 7
        v = (m / 8.0) * 2.0 ** -6 if e == 0 else (1 + m / 8.0) * 2.0 ** (e - 7)
        t[b] = -v if s else v
    return t
T = tab()

def run(a8, sa, b8, sb):
    dev = [torch.from_numpy(np.ascontiguousarray(t)).cuda() for t in (a8, sa, b8, sb)]
    d = torch.zeros((32, 32), dtype=torch.float32, device="cuda")
    _lib.check(lib.xmc_mx8_probe(*[C.c_void_p(t.data_ptr()) for t in dev], C.c_void_p(d.data_ptr()), None), "probe")
    return d.cpu().numpy().astype(np.float64)

gen = np.random.default_rng(0)
one = np.full((32, 2), 127, np.uint8)
a8 = gen.integers(0x30, 0x48, size=(32, 64), dtype=np.uint8)       # values around 1
b8 = gen.integers(0x30, 0x48, size=(32, 64), dtype=np.uint8)
want = T[a8] @ T[b8].T
got = run(a8, one, b8, one)
print("E1 unit scales: max rel err", np.abs(got - want).max() / np.abs(want).max())
# E2: A = ones (0x38 = 1.0), B = ones; scale of A at (row r, block h) = 128 (x2) -> which outputs change?
ones = np.full((32, 64), 0x38, np.uint8)
base = run(ones, one, ones, one)
print("base (expect 64):", np.unique(base))
for (r, h) in [(0, 0), (0, 1), (1, 0), (5, 1), (31, 0)]:
    sa = one.copy(); sa[r, h] = 128
    d = run(ones, sa, ones, one)
    ch = np.argwhere(d != base)
    rows, cols = np.unique(ch[:, 0]), np.unique(ch[:, 1])
    print(f"sa[{r},{h}]=128 -> changed rows {rows.tolist()[:8]} (n={len(rows)}), cols n={len(cols)}, values {np.unique(d[d != base])}")
for (r, h) in [(0, 0), (0, 1), (3, 0), (7, 1)]:
    sb = one.copy(); sb[r, h] = 128
    d = run(ones, one, ones, sb)
    ch = np.argwhere(d != base)
    rows, cols = np.unique(ch[:, 0]), np.unique(ch[:, 1])
    print(f"sb[{r},{h}]=128 -> changed cols {cols.tolist()[:8]} (n={len(cols)}), rows n={len(rows)}, values {np.unique(d[d != base])}")
# E3: which k positions does lane half h cover? A row 0 has a single 1.0 at k, B all ones with block scales 1 / 2
for k in (0, 5, 15, 16, 31, 32, 40, 63):
    a = np.zeros((32, 64), np.uint8); a[0, k] = 0x38
    sb = one.copy(); sb[:, 1] = 128                      # B block 1 scaled x2
    d = run(a, one, ones, sb)
    print(f"A[0,{k}]=1, B block1 x2 -> D[0,0] = {d[0, 0]}")

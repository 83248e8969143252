"""ctypes binding of ``libxmc_io.so`` (csrc_host/xmc_io.c): CRC-32C, PNG un-filter, bilinear resize.  Host-side
input-pipeline helpers only -- nothing here touches the GPU or the training step."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(_HERE, "libxmc_io.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} not found: build it with `make -C xmcgan_image_generation_amd/csrc_host` "
                               "(or __graft_entry__.build())")
        lib = C.CDLL(LIB_PATH)
        lib.xmc_crc32c.argtypes = [C.c_void_p, C.c_size_t]
        lib.xmc_crc32c.restype = C.c_uint32
        lib.xmc_crc32c_table.argtypes = [C.c_void_p, C.c_size_t]
        lib.xmc_crc32c_table.restype = C.c_uint32
        lib.xmc_crc32_ieee.argtypes = [C.c_void_p, C.c_size_t]
        lib.xmc_crc32_ieee.restype = C.c_uint32
        lib.xmc_masked_crc32c.argtypes = [C.c_void_p, C.c_size_t]
        lib.xmc_masked_crc32c.restype = C.c_uint32
        lib.xmc_png_unfilter.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
        lib.xmc_png_unfilter.restype = C.c_int
        lib.xmc_resize_bilinear_rgb.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
        lib.xmc_resize_bilinear_rgb.restype = C.c_int
        lib.xmc_png_info.argtypes = [C.c_void_p, C.c_int64] + [C.POINTER(C.c_int32)] * 4
        lib.xmc_png_info.restype = C.c_int
        lib.xmc_png_decode.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32]
        lib.xmc_png_decode.restype = C.c_int
        lib.xmc_inflate_zlib.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
        lib.xmc_inflate_zlib.restype = C.c_int
        lib.xmc_inflate_out_slack.restype = C.c_int32
        assert lib.xmc_io_abi_version() == 4
        _lib = lib
    return _lib


def _buf(b):
    """address of a bytes-like object without copying"""
    a = np.frombuffer(b, dtype=np.uint8)
    return a, a.ctypes.data


def crc32c(data) -> int:
    a, p = _buf(data)
    return int(load().xmc_crc32c(p, a.size))


def crc32c_table(data) -> int:
    """the portable slice-by-8 path (xmc_crc32c takes the SSE4.2 instruction where the CPU has it)"""
    a, p = _buf(data)
    return int(load().xmc_crc32c_table(p, a.size))


def crc32_ieee(data) -> int:
    """zlib.crc32 of ``data`` (the PNG chunk checksum), by carry-less multiplication where the CPU has it"""
    a, p = _buf(data)
    return int(load().xmc_crc32_ieee(p, a.size))


def masked_crc32c(data) -> int:
    a, p = _buf(data)
    return int(load().xmc_masked_crc32c(p, a.size))


def png_unfilter(raw: bytes, h: int, rowbytes: int, bpp: int) -> np.ndarray:
    a, p = _buf(raw)
    if a.size != h * (rowbytes + 1):
        raise ValueError("PNG: inflated size does not match the header")
    out = np.empty((h, rowbytes), np.uint8)
    if load().xmc_png_unfilter(p, out.ctypes.data, h, rowbytes, bpp) != 0:
        raise ValueError("PNG: unknown filter type")
    return out


def inflate_zlib(data, size: int) -> np.ndarray:
    """zlib stream -> uint8 (size,) with csrc_host/xmc_inflate.c (one-shot decoder; raises on malformed input or a size that
    does not match)"""
    a, p = _buf(data)
    lib = load()
    out = np.empty((size + int(lib.xmc_inflate_out_slack()),), np.uint8)
    rc = lib.xmc_inflate_zlib(p, a.size, out.ctypes.data, size)
    if rc != 0:
        raise ValueError(f"inflate: {'size mismatch / truncated stream' if rc == -2 else 'malformed stream'} ({rc})")
    return out[:size]


def png_decode(data, verify_crc: bool = True, use_zlib: bool = False):
    """whole-image PNG decode in C (chunk walk, CRC, inflate, un-filter; the GIL is released for the whole call) ->
    (uint8 (h, w, channels), colour type), or None when the image needs the Python path (palette, 16-bit, interlaced).
    ``use_zlib``: inflate with zlib instead of xmc_inflate.c (A/B and tests)."""
    a, p = _buf(data)
    lib = load()
    w, h, ch, ct = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
    rc = lib.xmc_png_info(p, a.size, C.byref(w), C.byref(h), C.byref(ch), C.byref(ct))
    if rc == 1:
        return None
    if rc == -2:
        raise ValueError(f"PNG: header claims {w.value} x {h.value} pixels -- larger than the decoder's 1 GiB cap")
    if rc != 0:
        raise ValueError("not a PNG")
    px = np.empty((h.value, w.value, ch.value), np.uint8)
    scratch = np.empty((h.value * (w.value * ch.value + 1) + int(lib.xmc_inflate_out_slack()),), np.uint8)
    rc = lib.xmc_png_decode(p, a.size, px.ctypes.data, scratch.ctypes.data, int(bool(verify_crc)) | (2 if use_zlib else 0))
    if rc != 0:
        raise ValueError({-5: "PNG: chunk crc mismatch", -7: "PNG: inflated size does not match the header",
                          -8: "PNG: unknown filter type"}.get(rc, f"PNG: malformed stream ({rc})"))
    return px, ct.value


def resize_bilinear_rgb(img_u8: np.ndarray, size: int, flip: bool = False) -> np.ndarray:
    """uint8 (H, W, 3) -> float32 (size, size, 3) in [0, 1] (tf.image.resize bilinear, half-pixel centres)."""
    img_u8 = np.ascontiguousarray(img_u8, dtype=np.uint8)
    assert img_u8.ndim == 3 and img_u8.shape[2] == 3
    out = np.empty((size, size, 3), np.float32)
    rc = load().xmc_resize_bilinear_rgb(img_u8.ctypes.data, img_u8.shape[0], img_u8.shape[1], out.ctypes.data, size, size, int(flip))
    if rc != 0:
        raise MemoryError(f"xmc_resize_bilinear_rgb: rc={rc}")
    return out

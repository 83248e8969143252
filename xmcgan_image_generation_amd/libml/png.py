"""Minimal PNG codec for the dataset's images (``tf.io.encode_png`` / ``tf.image.decode_png(channels=3)``,
``preprocess_data.py:80``, ``coco_dataset.py:97-98``): 8-bit, non-interlaced, colour types 0 (grey), 2 (RGB),
3 (palette), 4 (grey + alpha), 6 (RGBA) -> uint8 (H, W, 3).  zlib does the inflate, the C helper the scanline
un-filter (``csrc_host/xmc_io.c``)."""
from __future__ import annotations

import struct
import zlib

import numpy as np

from . import _io

_SIG = b"\x89PNG\r\n\x1a\n"
_CHANNELS = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}


def decode_rgb(data: bytes) -> np.ndarray:
    if data[:8] != _SIG:
        raise ValueError("not a PNG")
    fast = _io.png_decode(data)                  # one C call: chunk walk + CRC + inflate + un-filter (no Python in between)
    if fast is not None:
        px, ctype = fast
        if ctype == 2:
            return px
        if ctype == 6:
            return np.ascontiguousarray(px[..., :3])
        return np.repeat(px[..., :1], 3, axis=2)        # grey (+ alpha)
    return _decode_rgb_py(data)


def _decode_rgb_py(data: bytes) -> np.ndarray:
    """the Python path (zlib module + C un-filter): palette images, and the reference the C decoder is tested against"""
    pos, idat, plte, ihdr = 8, [], None, None
    while pos < len(data):
        ln, typ = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + ln]
        if zlib.crc32(typ + body) & 0xffffffff != struct.unpack(">I", data[pos + 8 + ln:pos + 12 + ln])[0]:
            raise ValueError("PNG: chunk crc mismatch")
        pos += 12 + ln
        if typ == b"IHDR":
            ihdr = struct.unpack(">IIBBBBB", body)
        elif typ == b"PLTE":
            plte = np.frombuffer(body, np.uint8).reshape(-1, 3)
        elif typ == b"IDAT":
            idat.append(body)
        elif typ == b"IEND":
            break
    w, h, depth, ctype, _, _, interlace = ihdr
    if depth != 8 or interlace != 0 or ctype not in _CHANNELS:
        raise ValueError(f"PNG: unsupported format (depth {depth}, colour type {ctype}, interlace {interlace})")
    ch = _CHANNELS[ctype]
    px = _io.png_unfilter(zlib.decompress(b"".join(idat)), h, w * ch, ch).reshape(h, w, ch)
    if ctype == 2:
        return px
    if ctype == 6:
        return np.ascontiguousarray(px[..., :3])
    if ctype == 3:
        return plte[px[..., 0]]
    return np.repeat(px[..., :1], 3, axis=2)            # grey (+ alpha)


def encode_rgb(img: np.ndarray, filter_types=None) -> bytes:
    """uint8 (H, W, 3) -> PNG bytes.  ``filter_types``: per-row PNG filter ids (default 0 = None); the
    filtered bytes are computed here in NumPy (test fixtures exercise all five filters of the decoder)."""
    img = np.ascontiguousarray(img, np.uint8)
    h, w, c = img.shape
    assert c == 3
    rows = img.reshape(h, w * 3).astype(np.int32)
    ft = np.zeros(h, np.int32) if filter_types is None else np.asarray(filter_types, np.int32)
    raw = bytearray()
    prev = np.zeros(w * 3, np.int32)
    for y in range(h):
        cur = rows[y]
        left = np.concatenate([np.zeros(3, np.int32), cur[:-3]])
        ul = np.concatenate([np.zeros(3, np.int32), prev[:-3]])
        if ft[y] == 0:
            f = cur
        elif ft[y] == 1:
            f = cur - left
        elif ft[y] == 2:
            f = cur - prev
        elif ft[y] == 3:
            f = cur - ((left + prev) >> 1)
        else:
            p = left + prev - ul
            pa, pb, pc = np.abs(p - left), np.abs(p - prev), np.abs(p - ul)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, ul))
            f = cur - pred
        raw.append(int(ft[y]))
        raw += (f & 0xff).astype(np.uint8).tobytes()
        prev = cur

    def chunk(typ, body):
        return struct.pack(">I", len(body)) + typ + body + struct.pack(">I", zlib.crc32(typ + body) & 0xffffffff)
    return _SIG + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + \
        chunk(b"IDAT", zlib.compress(bytes(raw), 6)) + chunk(b"IEND", b"")

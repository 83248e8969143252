"""TFRecord framing and ``tf.train.Example`` wire format without TensorFlow / protobuf.

The reference stores COCO as sharded TFRecords of ``tf.train.Example`` (``preprocess_data.py:76-96``) and reads
them back with ``tf.io.parse_single_example`` (``xmcgan/libml/coco_dataset.py:85-111``).  Both formats are tiny:

* TFRecord: ``[u64 length][u32 masked crc32c(length)][data][u32 masked crc32c(data)]`` per record, little endian,
  CRC-32C (Castagnoli) "masked" as ``rotr(crc, 15) + 0xa282ead8``;
* Example: protobuf ``Example{features=1: Features{feature=1: map<string, Feature>}}`` with
  ``Feature{bytes_list=1 | float_list=2 | int64_list=3}``, each list a ``repeated value=1`` (floats / int64 packed).

This module reads AND writes them (the writer produces test fixtures and lets a user build shards without TF).
"""
from __future__ import annotations

import struct
from typing import Dict, Iterator, List, Union

import numpy as np

from . import _io

Feature = Union[List[bytes], np.ndarray]


# ------------------------------------------------------------------------------------------- TFRecord framing
def read_records(path: str, verify_crc: bool = True) -> Iterator[bytes]:
    with open(path, "rb") as f:
        while True:
            head = f.read(12)
            if not head:
                return
            if len(head) != 12:
                raise IOError(f"{path}: truncated record header")
            (length,), (lcrc,) = struct.unpack("<Q", head[:8]), struct.unpack("<I", head[8:])
            if verify_crc and _io.masked_crc32c(head[:8]) != lcrc:
                raise IOError(f"{path}: corrupted record length (crc mismatch)")
            data = f.read(length)
            tail = f.read(4)
            if len(data) != length or len(tail) != 4:
                raise IOError(f"{path}: truncated record")
            if verify_crc and _io.masked_crc32c(data) != struct.unpack("<I", tail)[0]:
                raise IOError(f"{path}: corrupted record data (crc mismatch)")
            yield data


def write_records(path: str, records) -> None:
    with open(path, "wb") as f:
        for data in records:
            head = struct.pack("<Q", len(data))
            f.write(head)
            f.write(struct.pack("<I", _io.masked_crc32c(head)))
            f.write(data)
            f.write(struct.pack("<I", _io.masked_crc32c(data)))


# ---------------------------------------------------------------------------------------- protobuf wire format
def _varint(buf: bytes, pos: int):
    out = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7f) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def _fields(buf: bytes):
    """yield (field number, wire type, value) of one message; length-delimited values are memoryview slices"""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            v = buf[pos:pos + 4]
            pos += 4
        elif wt == 1:
            v = buf[pos:pos + 8]
            pos += 8
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        yield num, wt, v


def _parse_feature(buf: bytes) -> Feature:
    for num, wt, v in _fields(buf):
        if num == 1:                                    # BytesList
            return [bytes(x) for n2, _, x in _fields(v) if n2 == 1]
        if num == 2:                                    # FloatList: packed (one blob) or repeated fixed32
            parts = [bytes(x) for n2, _, x in _fields(v) if n2 == 1]
            return np.frombuffer(b"".join(parts), dtype="<f4")
        if num == 3:                                    # Int64List: packed varints or repeated varints
            vals = []
            for n2, wt2, x in _fields(v):
                if n2 != 1:
                    continue
                if wt2 == 0:
                    vals.append(x)
                else:
                    p = 0
                    while p < len(x):
                        y, p = _varint(x, p)
                        vals.append(y)
            a = np.array(vals, dtype=np.uint64).astype(np.int64)         # two's complement for negatives
            return a
    return []


def parse_example(data: bytes) -> Dict[str, Feature]:
    """serialized tf.train.Example -> {name: list of bytes | float32 array | int64 array}"""
    out: Dict[str, Feature] = {}
    data = memoryview(data)                             # nested messages as views: slicing bytes copied the ~0.9 MB image five times
    for num, _, feats in _fields(data):
        if num != 1:
            continue
        for n1, _, entry in _fields(feats):             # Features.feature map entries
            if n1 != 1:
                continue
            key, val = None, b""
            for n2, _, x in _fields(entry):
                if n2 == 1:
                    key = bytes(x).decode()
                elif n2 == 2:
                    val = x
            out[key] = _parse_feature(val)
    return out


def _enc_varint(v: int) -> bytes:
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7f
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _ld(num: int, payload: bytes) -> bytes:
    return _enc_varint((num << 3) | 2) + _enc_varint(len(payload)) + payload


def serialize_example(features: Dict[str, Feature]) -> bytes:
    """{name: list of bytes | float array | int array} -> serialized tf.train.Example (packed lists, as TF writes)"""
    entries = b""
    for key in sorted(features):
        v = features[key]
        if isinstance(v, (list, tuple)) and all(isinstance(x, (bytes, bytearray)) for x in v):
            feat = _ld(1, b"".join(_ld(1, bytes(x)) for x in v))
        else:
            a = np.asarray(v)
            if a.dtype.kind == "f":
                feat = _ld(2, _ld(1, a.astype("<f4").tobytes()))
            elif a.dtype.kind in "iu":
                feat = _ld(3, _ld(1, b"".join(_enc_varint(int(x)) for x in a.reshape(-1))))
            else:
                raise TypeError(f"feature {key}: unsupported dtype {a.dtype}")
        entries += _ld(1, _ld(1, key.encode()) + _ld(2, feat))
    return _ld(1, entries)

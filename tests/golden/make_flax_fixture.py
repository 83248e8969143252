#!/usr/bin/env python
"""Independent encoder of a Flax ``TrainState`` checkpoint: writes tests/golden/flax_state_small.msgpack.

Neither ``flax`` nor a reference checkpoint file exists in this image, so the product's reader / writer
(xmcgan_image_generation_amd/utils/checkpoint.py, built on the ``msgpack`` library) is pinned against a SECOND
implementation of the same public formats that shares no code with it:

  * the msgpack wire format, written here byte by byte from the msgpack specification (fixmap / map16, fixstr / str8,
    positive fixint / uint8..32, bin8 / bin16 / bin32, fixarray, ext8 / ext16 / ext32);
  * the Flax convention on top of it (flax/serialization.py of flax 0.3.3, the release the reference pins in
    requirements.txt:13): a state dict is a map with str keys; an ndarray leaf is extension type 1 whose payload is the
    msgpack array [shape (array of ints), dtype name (str), raw little-endian bytes (bin)]; Python ints stay native;
    ``TrainState`` serialises field by field (train_utils.py:42-50), an ``Optimizer`` as {"target", "state": {"step",
    "param_states"}} with Adam's per-parameter {"grad_ema", "grad_sq_ema"}.

The tree below has the reference's structure and Flax-layout shapes (conv HWIO, dense (in, out)) at toy sizes.
usage: python tests/golden/make_flax_fixture.py     (deterministic: NumPy default_rng(2024))"""
import os
import struct

import numpy as np


def enc_int(v):
    assert 0 <= v < 2 ** 32
    if v < 128:
        return bytes([v])
    if v < 256:
        return b"\xcc" + struct.pack(">B", v)
    if v < 65536:
        return b"\xcd" + struct.pack(">H", v)
    return b"\xce" + struct.pack(">I", v)


def enc_str(s):
    b = s.encode()
    if len(b) < 32:
        return bytes([0xA0 | len(b)]) + b
    assert len(b) < 256
    return b"\xd9" + bytes([len(b)]) + b


def enc_bin(b):
    if len(b) < 256:
        return b"\xc4" + bytes([len(b)]) + b
    if len(b) < 65536:
        return b"\xc5" + struct.pack(">H", len(b)) + b
    return b"\xc6" + struct.pack(">I", len(b)) + b


def enc_ext(code, payload):
    n = len(payload)
    if n in (1, 2, 4, 8, 16):
        return bytes([{1: 0xD4, 2: 0xD5, 4: 0xD6, 8: 0xD7, 16: 0xD8}[n], code]) + payload
    if n < 256:
        return b"\xc7" + bytes([n, code]) + payload
    if n < 65536:
        return b"\xc8" + struct.pack(">H", n) + bytes([code]) + payload
    return b"\xc9" + struct.pack(">I", n) + bytes([code]) + payload


def enc_ndarray(a):
    a = np.asarray(a)
    shape = bytes([0x90 | a.ndim]) + b"".join(enc_int(int(d)) for d in a.shape)
    return enc_ext(1, b"\x93" + shape + enc_str(a.dtype.name) + enc_bin(a.astype(a.dtype.newbyteorder("<")).tobytes()))


def enc(tree):
    if isinstance(tree, dict):
        n = len(tree)
        head = bytes([0x80 | n]) if n < 16 else b"\xde" + struct.pack(">H", n)
        return head + b"".join(enc_str(k) + enc(v) for k, v in tree.items())
    if isinstance(tree, np.ndarray):
        return enc_ndarray(tree)
    if isinstance(tree, int):
        return enc_int(tree)
    raise TypeError(type(tree))


def build_tree():
    g = np.random.default_rng(2024)
    f32 = lambda *s: g.standard_normal(s).astype(np.float32)

    def adam(params):
        if isinstance(params, dict):
            return {k: adam(v) for k, v in params.items()}
        return {"grad_ema": f32(*params.shape), "grad_sq_ema": np.abs(f32(*params.shape))}

    def optimizer(params, step):
        return {"target": params, "state": {"step": np.asarray(step, np.int32), "param_states": adam(params)}}
    g_params = {"Dense_0": {"kernel": f32(6, 4), "bias": f32(4)},
                "GenBlock_0": {"Conv_0": {"kernel": f32(3, 3, 4, 5), "bias": f32(5)},
                               "ConditionalBatchNorm_0": {"Dense_0": {"kernel": f32(4, 4), "bias": f32(4)}}}}
    d_params = {"DiscBlock_0": {"SpectralConv_0": {"kernel": f32(3, 3, 5, 7), "bias": f32(7)}},
                "SpectralDense_0": {"kernel": f32(7, 1), "bias": f32(1)}}
    return {"step": 12345,
            "g_optimizer": optimizer(g_params, 12345),
            "d_optimizer": optimizer(d_params, 24690),
            "generator_state": {"batch_stats": {"GenBlock_0": {"ConditionalBatchNorm_0": {"BatchNorm_0": {"mean": f32(4), "var": np.abs(f32(4))}}}}},
            "discriminator_state": {"spectral_norm_stats": {"DiscBlock_0": {"SpectralConv_0": {"u0": f32(1, 7)}},
                                                            "SpectralDense_0": {"u0": f32(1, 1)}}},
            "ema_params": {k: ({kk: (vv if not isinstance(vv, dict) else {k3: v3 for k3, v3 in vv.items()}) for kk, vv in v.items()})
                           for k, v in g_params.items()}}


if __name__ == "__main__":
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "flax_state_small.msgpack")
    data = enc(build_tree())
    with open(out, "wb") as f:
        f.write(data)
    print(out, len(data), "bytes")

"""Checkpoint layout compatibility with the reference.

The reference stores ``flax.serialization.to_bytes(TrainState)`` (``xmcgan/utils/task_manager.py:63-67``,
``train_utils.py:372-375,459``): a msgpack map of the state dict

    {"step", "g_optimizer": {"target": <params>, "state": {"step", "param_states": <params-shaped tree of
     {"grad_ema", "grad_sq_ema"}>}}, "d_optimizer": {...}, "generator_state": {"batch_stats": ...},
     "discriminator_state": {"spectral_norm_stats": ...}, "ema_params": <params>}

whose array leaves are msgpack extension type 1 = ``packb((shape, dtype.name, raw bytes))`` (numpy scalars: type
3; arrays above 2**30 bytes are split into a ``__msgpack_chunked_array__`` map -- no XMC-GAN leaf is that large,
the reader still accepts it).  ``flax`` is not installed here and the reference's tests hold no checkpoint file,
so this is a restatement of that public format (**unpinned**, like the oracle): the tests pin the byte layout
of small hand-built examples and a full round trip.  Kernels are stored in the Flax layouts the arena exposes as
views (conv HWIO, dense (in, out)), all leaves float32 except the integer step counters.
"""
from __future__ import annotations

from typing import Any, Dict

import msgpack
import numpy as np
import torch

from .. import synthetic as syn
from ..nets import xmc_net

_EXT_NDARRAY, _EXT_NATIVE_COMPLEX, _EXT_NPSCALAR = 1, 2, 3
_CHUNK_KEY = "__msgpack_chunked_array__"


# ------------------------------------------------------------------------------------------ msgpack layer
def _ndarray_to_bytes(arr: np.ndarray) -> bytes:
    arr = np.asarray(arr)
    if arr.dtype.hasobject or arr.dtype.isalignedstruct:
        raise ValueError("object and structured dtypes are not serialisable")
    return msgpack.packb((tuple(arr.shape), arr.dtype.name, arr.tobytes()), use_bin_type=True)


def _ndarray_from_bytes(data: bytes) -> np.ndarray:
    shape, dtype_name, buffer = msgpack.unpackb(data, raw=False)
    return np.frombuffer(buffer, dtype=np.dtype(dtype_name), count=-1).reshape(tuple(shape)).copy()


def _ext_pack(x):
    if isinstance(x, torch.Tensor):
        x = x.detach().cpu().numpy()
    if isinstance(x, np.ndarray):
        return msgpack.ExtType(_EXT_NDARRAY, _ndarray_to_bytes(x))
    if isinstance(x, np.generic):
        return msgpack.ExtType(_EXT_NPSCALAR, _ndarray_to_bytes(np.asarray(x)))
    if isinstance(x, complex):
        return msgpack.ExtType(_EXT_NATIVE_COMPLEX, msgpack.packb((x.real, x.imag)))
    raise TypeError(f"cannot serialise {type(x)}")


def _ext_unpack(code, data):
    if code == _EXT_NDARRAY:
        return _ndarray_from_bytes(data)
    if code == _EXT_NPSCALAR:
        return _ndarray_from_bytes(data)[()]
    if code == _EXT_NATIVE_COMPLEX:
        re, im = msgpack.unpackb(data)
        return complex(re, im)
    return msgpack.ExtType(code, data)


def _unchunk(tree):
    if isinstance(tree, dict):
        if tree.get(_CHUNK_KEY):
            shape = tuple(tree["shape"][str(i)] for i in range(len(tree["shape"])))
            chunks = [tree["chunks"][str(i)] for i in range(len(tree["chunks"]))]
            return np.concatenate([np.asarray(c).reshape(-1) for c in chunks]).reshape(shape)
        return {k: _unchunk(v) for k, v in tree.items()}
    return tree


def msgpack_serialize(tree: Dict[str, Any]) -> bytes:
    """flax.serialization.msgpack_serialize: nested dicts with str keys, ndarray / numpy-scalar / int leaves."""
    return msgpack.packb(tree, default=_ext_pack, strict_types=True)


def msgpack_restore(data: bytes) -> Dict[str, Any]:
    return _unchunk(msgpack.unpackb(data, ext_hook=_ext_unpack, raw=False, strict_map_key=False))


# ------------------------------------------------------------------------------------------ TrainState <-> dict
def _np_tree(tree):
    return syn.tree_map(lambda t: t.detach().float().cpu().numpy() if torch.is_tensor(t) else np.asarray(t), dict(tree))


def _opt_dict(opt):
    a = opt.arena
    m, v = _np_tree(a.tree(a.m)), _np_tree(a.tree(a.v))
    states = syn.tree_map(lambda gm, gv: {"grad_ema": gm, "grad_sq_ema": gv}, m, v)
    return {"target": _np_tree(a.tree()), "state": {"step": np.asarray(a.opt_step, np.int32), "param_states": states}}


def to_state_dict(state) -> Dict[str, Any]:
    """TrainState -> the nested dict ``flax.serialization.to_state_dict`` produces for the reference's TrainState."""
    if getattr(state, "pending", None) is not None:
        raise ValueError("a deferred discriminator update is pending; checkpoint after train_step returns")
    return {"step": int(state.step),
            "g_optimizer": _opt_dict(state.g_optimizer),
            "d_optimizer": _opt_dict(state.d_optimizer),
            "generator_state": _np_tree(state.generator_state),
            "discriminator_state": _np_tree(state.discriminator_state),
            "ema_params": _np_tree(state.g_optimizer.arena.tree(state.ema_buffer))}


def _load_opt(opt, d):
    a = opt.arena
    a.load_flax(d["target"])
    ps = d["state"]["param_states"]
    a.load_flax(_restructure(ps, "grad_ema"), a.m)
    a.load_flax(_restructure(ps, "grad_sq_ema"), a.v)
    a.opt_step = int(np.asarray(d["state"]["step"]))
    opt.target = a.tree()


def _restructure(param_states, field):
    """params-shaped tree of {"grad_ema", "grad_sq_ema"} -> params-shaped tree of one of the two."""
    if isinstance(param_states, dict) and field in param_states and not isinstance(param_states[field], dict):
        return param_states[field]
    return {k: _restructure(v, field) for k, v in param_states.items()}


def from_state_dict(state, d: Dict[str, Any]):
    """Install a (reference-layout) state dict into an existing TrainState of the same configuration."""
    _load_opt(state.g_optimizer, d["g_optimizer"])
    _load_opt(state.d_optimizer, d["d_optimizer"])
    state.g_optimizer.arena.load_flax(d["ema_params"], state.ema_buffer)
    ops = state.g_optimizer.arena.ops
    return state.replace(step=int(np.asarray(d["step"])),
                         generator_state={k: xmc_net._tree_to_dev(ops, v) for k, v in d["generator_state"].items()},
                         discriminator_state={k: xmc_net._tree_to_dev(ops, v) for k, v in d["discriminator_state"].items()},
                         ema_params=state.g_optimizer.arena.tree(state.ema_buffer), pending=None)


def to_bytes(state) -> bytes:
    return msgpack_serialize(to_state_dict(state))


def from_bytes(state, data: bytes):
    return from_state_dict(state, msgpack_restore(data))


def save(path: str, state) -> None:
    with open(path, "wb") as f:
        f.write(to_bytes(state))


def restore(path: str, state):
    with open(path, "rb") as f:
        return from_bytes(state, f.read())

"""Image augmentation of the input pipeline (reference ``xmcgan/libml/augmentation.py``): ``augment`` (:25-70),
``augment_shift`` (:73-89, reflect-pad by w and random crop), ``augment_zoom_crop`` (:92-117, resize by 9/8 and
random crop), random left-right flip.  NumPy on the host; images are float32 NHWC batches.  The reference draws its
randomness from TF stateless RNGs; here a seed (int or ``np.random.Generator``) gives the same determinism contract
the reference's tests check (same seed -> same output, different seeds -> different outputs)."""
from __future__ import annotations

import numpy as np

_SHIFT = "shift"
_ZOOM_CROP = "zoom_crop"


def _rng(seed):
    return seed if isinstance(seed, np.random.Generator) else np.random.default_rng(seed)


def _random_crop(y, shape, rng):
    n, h, w, c = shape
    oy = int(rng.integers(0, y.shape[1] - h + 1))
    ox = int(rng.integers(0, y.shape[2] - w + 1))
    return y[:, oy:oy + h, ox:ox + w, :]


def _resize_bilinear(x, size):
    """half-pixel-centre bilinear resize of a float NHWC batch (tf.image.resize, antialias=False)"""
    n, h, w, c = x.shape

    def grid(src, dst):
        f = (np.arange(dst, dtype=np.float64) + 0.5) * (src / dst) - 0.5
        lo = np.floor(f)
        t = (f - lo).astype(x.dtype)
        i0 = np.clip(lo, 0, src - 1).astype(np.int64)
        i1 = np.clip(np.ceil(f), 0, src - 1).astype(np.int64)
        return i0, i1, t
    y0, y1, ty = grid(h, size)
    x0, x1, tx = grid(w, size)
    top = x[:, y0][:, :, x0] + (x[:, y0][:, :, x1] - x[:, y0][:, :, x0]) * tx[None, None, :, None]
    bot = x[:, y1][:, :, x0] + (x[:, y1][:, :, x1] - x[:, y1][:, :, x0]) * tx[None, None, :, None]
    return top + (bot - top) * ty[None, :, None, None]


def augment_shift(x, w: int = 4, seed=None):
    """Randomly translates the image by up to w pixels (reflect padding) -- augmentation.py:73-89."""
    y = np.pad(x, [(0, 0), (w, w), (w, w), (0, 0)], mode="reflect")
    return np.ascontiguousarray(_random_crop(y, x.shape, _rng(seed)))


def augment_zoom_crop(x, resize_method: str = "bilinear", zoom_ratio: float = 1.125, seed=None):
    """Randomly zooms and crops the image -- augmentation.py:92-117."""
    if resize_method not in ("nearest", "bilinear"):
        raise NotImplementedError(f"{resize_method} is not supported.")
    new_size = int(float(x.shape[1]) * zoom_ratio)
    if resize_method == "bilinear":
        y = _resize_bilinear(x, new_size)
    else:
        idx = np.floor((np.arange(new_size) + 0.5) * (x.shape[1] / new_size)).astype(np.int64)
        y = x[:, idx][:, :, idx]
    return np.ascontiguousarray(_random_crop(y, x.shape, _rng(seed)))


def augment(x, method: str = _SHIFT, random_flip: bool = True, resize_method: str = "bilinear", seed=None, **kwargs):
    """Randomly augments the input image batch -- augmentation.py:25-70."""
    rng = _rng(seed)
    rng_shift, rng_zoom, rng_flip = (np.random.default_rng(s) for s in rng.integers(0, 2 ** 63 - 1, size=3))
    if method == _SHIFT:
        x = augment_shift(x, seed=rng_shift, **kwargs)
    elif method == _ZOOM_CROP:
        x = augment_zoom_crop(x, seed=rng_zoom, resize_method=resize_method, **kwargs)
    else:
        raise NotImplementedError(f"{method} is not supported for data augmentation.")
    if random_flip and rng_flip.random() < 0.5:
        x = np.ascontiguousarray(x[:, :, ::-1, :])
    return x

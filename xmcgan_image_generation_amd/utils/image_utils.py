"""Image grids for the sampling path (reference ``xmcgan/utils/image_utils.py:23-38``)."""
from __future__ import annotations

import math

import torch


def make_grid(samples: torch.Tensor, show_num: int = 64) -> torch.Tensor:
    """(B, H, W, C) -> (h_num * H, w_num * W, C): the first ``h_num * w_num`` samples tiled row-major, with
    ``h_num = floor(sqrt(n))``, ``w_num = floor(n / h_num)`` and ``n = min(show_num, B)`` (image_utils.py:26-37)."""
    b, h, w, c = samples.shape
    n = min(int(show_num), b)
    h_num = int(math.sqrt(n))
    w_num = n // h_num
    tiles = samples[:h_num * w_num].reshape(h_num, w_num, h, w, c)
    return tiles.permute(0, 2, 1, 3, 4).reshape(h_num * h, w_num * w, c)

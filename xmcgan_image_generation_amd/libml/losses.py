"""GAN hinge loss on the HIP operator table (reference ``xmcgan/libml/losses.py:30-35``)."""
from __future__ import annotations


def hinge_loss(ops, logit, batch, d_loss_acc, g_loss_acc):
    """logit (2B,) float32 = [real; fake].  Adds ``mean(relu(1-real)+relu(1+fake))`` into
    ``d_loss_acc`` and ``-mean(fake)`` into ``g_loss_acc``; returns (d hinge_d / d logit,
    d hinge_g / d logit), each (2B,)."""
    return ops.hinge(logit, batch, d_loss_acc, g_loss_acc)

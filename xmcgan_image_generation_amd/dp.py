"""Data-parallel replicas: gradient averaging over RCCL / xGMI.

The reference's only parallelism is ``jax.pmap`` replicas with ``lax.pmean`` of the G and D gradient
pytrees (``xmcgan/xmc_gan.py:170-171,251``; ``train_utils.py:378-388``): BatchNorm statistics,
spectral-norm vectors and contrastive negatives stay per-replica.  Here: one process
per GPU, ``torch.distributed`` (backend "nccl" == RCCL on ROCm), and each network's gradients are ONE
flat float32 arena, all-reduced in a few large buckets (xGMI is point-to-point, ~153 GB/s per link:
large messages, few launches) issued from a SIDE stream so the exchange of the discriminator
gradients overlaps the generator's backward pass.  The sum is turned into the mean by the 1/world
factor folded into the Adam kernel (``grad_scale``).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class GradSync:
    """``transport="bf16"`` (optional, off by default: it changes the numerics the parity tests pin): the gradients travel
    as bfloat16 -- half the bytes on the xGMI links -- and are summed by RCCL in bfloat16; the
    float32 arena is rewritten with the widened sum when the exchange is waited for.  ``transport="float32"`` is the
    reference's ``lax.pmean`` of float32 gradients.

    ``schedule``: WHEN the half steps issue their exchanges (same sums, same parameters -- bit for bit -- either way):
      "overlapped" (default)  sliced from inside the backward passes on the side stream: D's exchange of train_d under the next
                              generator forward, D's of train_g_d under G's backward pass, G's in three buckets behind its slices;
      "exclusive"             each arena in one piece AFTER the backward passes of its half step have finished, waited for before
                              the optimiser kernels start: nothing runs beside the RCCL kernels.  Cost = bytes / link bandwidth,
                              with no contention term (a resident neighbour costs the convolution kernels 18-28 %, profiles/
                              r05_cu_contention.txt): the schedule that cannot lose to the overlap going wrong on real xGMI links."""

    def __init__(self, bucket_elems=32 * 1024 * 1024, group=None, transport="float32", schedule="overlapped"):
        if not dist.is_initialized():
            raise RuntimeError("GradSync needs torch.distributed to be initialised")
        if transport not in ("float32", "bf16"):
            raise ValueError("transport is 'float32' or 'bf16'")
        if schedule not in ("overlapped", "exclusive"):
            raise ValueError("schedule is 'overlapped' or 'exclusive'")
        self.schedule = schedule
        self.group = group
        self.world = dist.get_world_size(group)
        self.bucket = int(bucket_elems)
        self.transport = transport
        self._works = {}
        self._half = {}             # tag -> [(float32 slice, bf16 copy)] to widen back in wait()
        self._side = None

    @property
    def exclusive(self) -> bool:
        return self.schedule == "exclusive"

    def _side_stream(self, device):
        if device.type != "cuda":
            return None
        if self._side is None:
            self._side = torch.cuda.Stream(device=device)
        return self._side

    def all_reduce(self, flat: torch.Tensor, tag: str, append: bool = False) -> float:
        """Start the (asynchronous) sum all-reduce of ``flat``; returns the scale (1/world) the
        consumer must apply.  ``flat`` must not be written until ``wait(tag)``.  ``append``: add this exchange to the
        ones already in flight under ``tag`` (bucketed exchange of one arena, issued slice by slice as the backward
        pass completes them)."""
        works = list(self._works.get(tag, [])) if append else []
        side = self._side_stream(flat.device)
        if not append:
            self._half[tag] = []
        if self.transport == "bf16":
            # precision: RCCL then SUMS in bfloat16 across the ranks (8 mantissa bits: an 8-rank sum carries ~3 more bits of
            # rounding than the reference's float32 pmean) -- the price of half the xGMI bytes; off by default
            half = flat.to(torch.bfloat16)                       # one cast pass on the producer stream
            if side is not None:
                half.record_stream(side)                         # allocated on the producer stream, reduced on the side stream
            self._half.setdefault(tag, []).append((flat, half))
            flat = half
        chunks = [flat[i:i + self.bucket] for i in range(0, flat.numel(), self.bucket)]
        if side is not None:
            side.wait_stream(torch.cuda.current_stream())        # gradients are complete
            with torch.cuda.stream(side):
                for c in chunks:
                    works.append(dist.all_reduce(c, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            for c in chunks:
                works.append(dist.all_reduce(c, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        self._works[tag] = works
        return 1.0 / self.world

    def wait(self, tag: str):
        """Make the current stream (GPU) / the host (gloo) wait for the exchange tagged ``tag``."""
        for w in self._works.pop(tag, []):
            w.wait()
        for dst, half in self._half.pop(tag, []):
            if half.is_cuda:
                half.record_stream(torch.cuda.current_stream())  # ... and widened on the consumer's stream
            dst.copy_(half)                                      # widen the summed bf16 gradients back into the arena

    def mean_metrics(self, metrics: dict) -> dict:
        """TrainMetrics.gather_from_model_output (xmc_gan.py:185-190): mean over replicas."""
        keys = sorted(metrics)
        v = torch.stack([metrics[k].reshape(()).float() for k in keys])
        dist.all_reduce(v, group=self.group)
        v = v / self.world
        return {k: v[i] for i, k in enumerate(keys)}

#!/usr/bin/env python
"""RCCL all-reduce launch + copy cost on ONE GPU (world size 1, backend "nccl" = RCCL) for the message sizes of the
gradient exchange: the fixed per-collective cost that goes into the MODELLED multi-GPU curve of DESIGN.md (a 1-GPU box
cannot measure xGMI; SURVEY.md 8(e) asks for the model to be labelled as such).
usage: python tools/rccl_loopback_sweep.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29571")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    for mb in (1, 8, 32, 128, 314, 352):
        x = torch.ones((mb * 1024 * 1024 // 4,), device="cuda")
        for _ in range(3):
            dist.all_reduce(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            dist.all_reduce(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"all_reduce float32 {mb:4d} MB, world 1: {dt * 1e3:7.3f} ms per call ({mb / 1024 / dt:7.1f} GB/s through the kernel)")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""hipGraph capture next to ProcessGroupNCCL's watchdog thread (the situation of bench.py under torchrun).

An eager RCCL all-reduce is issued, the device is synchronised, and a capture that lasts longer than the watchdog's
100 ms polling interval starts at once: the watchdog's hipEventQuery of the finished collective then lands INSIDE
the capture.  With train_utils.CAPTURE_ERROR_MODE ("thread_local") that is legal; with "global" the query fails with
hipErrorStreamCaptureUnsupported and the watchdog aborts the process.
    RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29571 python tools/capture_vs_watchdog.py [--mode global]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from xmcgan_image_generation_amd import train_utils  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default=train_utils.CAPTURE_ERROR_MODE)
    ap.add_argument("--rounds", type=int, default=4)
    a = ap.parse_args()
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    x = torch.ones(1 << 20, device="cuda")
    y = torch.zeros(1 << 20, device="cuda")
    for r in range(a.rounds):
        dist.all_reduce(x)                                   # an eager collective the watchdog will poll
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode=a.mode):
            for _ in range(8):
                time.sleep(0.05)                             # 0.4 s of capture: >= 3 watchdog polls inside
                y.add_(x)
            dist.all_reduce(y)                               # a captured collective, as in the training step
        g.replay()
        torch.cuda.synchronize()
        print(f"round {r}: capture ({a.mode}) + replay ok, y[0] = {float(y[0]):.1f}", flush=True)
    dist.destroy_process_group()
    print("capture vs watchdog OK")


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Decode throughput of the TFRecord / PNG input pipeline on the host cores of the GPU box.

Writes COCO-shaped synthetic shards (PNG images of COCO's typical 640 x 480, 5 captions x 17 x 768 float32 BERT
embeddings per example -- 261 KB of embeddings as in the reference's preprocess_data.py:76-96), then times
create_datasets' training iterator (parse_example + PNG decode + resize to 128 + flip + caption selection + batching)
for several decode-thread counts.  The C1 step consumes 2 x 56 images per 42 ms = ~2.7 k examples/s per GPU.
usage: python tools/bench_input_pipeline.py [--examples 256] [--workers 1,4,8,16]"""
import argparse
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from xmcgan_image_generation_amd.configs import coco_xmc  # noqa: E402
from xmcgan_image_generation_amd.libml import input_pipeline, png, tfrecord  # noqa: E402


def smooth_image(rng, h, w):
    """a compressible photo-like image: low-frequency colour field + mild noise (random noise would not deflate)"""
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.zeros((h, w, 3), np.float32)
    for c in range(3):
        fx, fy, ph = rng.uniform(0.005, 0.03, 2).tolist() + [rng.uniform(0, 6.28)]
        img[..., c] = 127 + 100 * np.sin(fx * xx + fy * yy + ph)
    img += rng.normal(0, 4, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--examples", type=int, default=192)
    ap.add_argument("--workers", default="1,4,8,16")
    ap.add_argument("--batches", type=int, default=6)
    ap.add_argument("--procs", default="0", help="comma list of decode PROCESS counts (each with --threads-per-proc threads)")
    ap.add_argument("--threads-per-proc", default="4", help="comma list")
    ap.add_argument("--shards", type=int, default=4)
    args = ap.parse_args()
    rng = np.random.default_rng(0)
    with tempfile.TemporaryDirectory() as d:
        t0 = time.perf_counter()
        nshard = args.shards
        raw = 0
        for s in range(nshard):
            recs = []
            for _ in range(args.examples // nshard):
                img = smooth_image(rng, 480, 640)
                data = png.encode_rgb(img, np.full(480, 4))         # Paeth rows, like a real encoder would mostly choose
                raw += len(data)
                emb = rng.standard_normal((5, 17, 768)).astype(np.float32)
                recs.append(tfrecord.serialize_example({
                    "image": [data], "image/filename": [b"x.jpg"], "caption/text": [b"a caption"] * 5,
                    "caption/embedding": emb.reshape(-1), "caption/max_len": rng.integers(4, 18, 5).astype(np.int64)}))
            tfrecord.write_records(os.path.join(d, f"coco2014_train.tfrecord-{s}-of-{nshard}"), recs)
            tfrecord.write_records(os.path.join(d, f"coco2014_validation.tfrecord-{s}-of-{nshard}"), recs[:2])
        quota = None
        try:                                            # the container's CPU quota (cgroup v2), not the host's core count, bounds the decode rate
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()
            quota = None if q == "max" else float(q) / float(per)
        except (OSError, ValueError):
            pass
        print(f"wrote {args.examples} examples, mean PNG {raw / args.examples / 1024:.0f} KiB, in {time.perf_counter() - t0:.1f} s; "
              f"host cores {len(os.sched_getaffinity(0))}, cgroup CPU quota {quota if quota is not None else 'none'} cores")
        cfg = coco_xmc.get_c1_config()
        cfg.update(data_dir=d + "/", coco_version="2014", shuffle_buffer_size=64, train_shuffle=True, eval_batch_size=2,
                   dataset="mscoco")
        per_batch = cfg.batch_size * cfg.d_step_per_g_step
        for w in [int(v) for v in args.workers.split(",")]:
            it, _, _ = input_pipeline.create_datasets(cfg, data_rng=1, workers=w, prefetch=2)
            next(it)                                              # thread start-up, first files open
            t0 = time.perf_counter()
            for _ in range(args.batches):
                next(it)
            dt = time.perf_counter() - t0
            print(f"decode workers {w:3d}: {args.batches * per_batch / dt:8.1f} examples/s "
                  f"({args.batches} batches of {per_batch})")
            del it
        for pr in [int(v) for v in args.procs.split(",") if int(v) > 0]:
            for tpp in [int(v) for v in str(args.threads_per_proc).split(",")]:
                it, _, _ = input_pipeline.create_datasets(cfg, data_rng=1, workers=tpp, procs=pr, prefetch=4)
                # the parent takes one batch from each worker process in turn, and the workers finish their batches at about the
                # same time: batches arrive in bursts of `pr`.  Warm up over two whole rounds and time WHOLE rounds (>= 5), or the
                # count depends on where in a burst the clock starts and stops (round 4's table under-read 16 x 2 that way).
                for _ in range(2 * pr + 2):
                    next(it)                                      # process start-up (spawn + imports), first round
                nb = pr * max(5, -(-args.batches // pr))
                t0 = time.perf_counter()
                for _ in range(nb):
                    next(it)
                dt = time.perf_counter() - t0
                print(f"decode processes {pr:3d} x {tpp} threads: {nb * per_batch / dt:8.1f} examples/s "
                      f"({nb} batches of {per_batch}, whole rounds)", flush=True)
                del it


if __name__ == "__main__":
    main()

"""Training / evaluation input pipeline (reference ``xmcgan/libml/input_pipeline.py:30-110``).

``create_datasets(config, data_rng)`` -> ``(train_iter, eval_iter, num_train_examples)``: each iterator yields the
per-device batch dict of the step -- leading dim ``per_device_batch * d_step_per_g_step`` for training
(input_pipeline.py:46-47) -- built by ``COCODataset`` from the reference's sharded TFRecords: shard ``rank`` of
``world`` reads files ``rank::world`` (one process per GPU), a shuffle buffer, endless repetition for training.
A background thread decodes ahead of the consumer (``prefetch`` batches) and, on a GPU, hands the batch over as
device tensors through pinned host buffers on a copy stream, so the step's inputs are resident in HBM when it starts.
"""
from __future__ import annotations

import os
import queue
import threading
import time
from typing import Dict, Iterator

import numpy as np

from . import tfrecord
from .coco_dataset import COCODataset


def _records(files, seed: int, shuffle: bool, shuffle_buffer: int, repeat: bool):
    """(global record index, serialized Example) in reading order: files shuffled per epoch, records through a shuffle
    buffer (tf.data's ``shuffle(buffer)`` semantics), endless when ``repeat``."""
    rng = np.random.default_rng(seed)
    index = 0
    while True:
        order = list(files)
        if shuffle:
            rng.shuffle(order)
        buf = []
        for path in order:
            for rec in tfrecord.read_records(path):
                buf.append((index, rec))
                index += 1
                if len(buf) >= max(1, shuffle_buffer if shuffle else 1):
                    yield buf.pop(int(rng.integers(0, len(buf))) if shuffle else 0)
        while buf:
            yield buf.pop(int(rng.integers(0, len(buf))) if shuffle else 0)
        if not repeat:
            return


def _examples(ds: COCODataset, files, seed, shuffle: bool, shuffle_buffer: int, repeat: bool, training: bool,
              workers: int = 1):
    """Decoded + augmented examples in record order.  ``seed`` is a sequence of ints (stream, rank): the per-example
    generator is ``default_rng([*seed, i])`` so replicas draw different z / flip / caption streams (the reference folds
    ``jax.host_id()`` into ``data_rng``, train_utils.py:333).  ``workers`` > 1: ``parse_example`` + ``preprocess`` (PNG
    inflate / un-filter / resize in C, GIL released) run on a thread pool, ``2 * workers`` records ahead, results kept
    in order (the reference maps with ``num_parallel_calls=AUTOTUNE``, base_dataset.py:69-72)."""
    seed = [int(v) for v in np.atleast_1d(seed)]
    recs = _records(files, int(np.random.SeedSequence(seed).generate_state(1)[0]), shuffle, shuffle_buffer, repeat)

    def decode(item):
        i, r = item
        return ds.preprocess(ds.parse_example(r), np.random.default_rng(seed + [i]), training)

    if workers <= 1:
        for item in recs:
            yield decode(item)
        return
    import collections
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=workers, thread_name_prefix="xmc-decode") as pool:
        pending = collections.deque()
        for item in recs:
            pending.append(pool.submit(decode, item))
            if len(pending) >= 2 * workers:
                yield pending.popleft().result()
        while pending:
            yield pending.popleft().result()


class SlotBatch(dict):
    """A batch whose arrays are views of a worker's shared-memory slot: valid until ``release()`` (which hands the slot back
    to the worker) -- the Prefetcher copies the arrays out (into pinned staging buffers / fresh arrays) and releases."""
    _release = None

    def release(self):
        if self._release is not None:
            self._release()
            self._release = None


_ERR = "__xmc_worker_error__"


def _mp_worker(q, free_q, ds_kw, files, seed, shuffle, shuffle_buffer, repeat, training, threads, batch, nslots):
    """child process of _batches_mp: decode its share of the shards, assemble whole batches and write them into a RING of
    ``nslots`` preallocated shared-memory slots (one segment per worker, created once: no per-batch segment creation, file-
    descriptor passing or first-touch page faults); only (slot index, non-array fields) travel through the queue."""
    shm = None
    try:
        from multiprocessing import shared_memory
        try:
            import torch
            torch.set_num_threads(1)
        except Exception:
            pass
        ds = COCODataset(**ds_kw)
        views = None
        stopped = False                              # the parent's stop token (None on free_q) has been consumed
        for b in _batches(_examples(ds, files, seed, shuffle, shuffle_buffer, repeat, training, threads), batch):
            arrays = {k: np.ascontiguousarray(v) for k, v in b.items() if isinstance(v, np.ndarray) and k != "image_aug"}
            others = {k: v for k, v in b.items() if not isinstance(v, np.ndarray)}
            # image_aug is a copy of image in this pipeline and is never read by the step (coco_dataset.py:138,156: SURVEY 8a);
            # the parent re-attaches it as an alias instead of shipping 22 MB per batch twice
            alias_aug = "image_aug" in b and "image" in b and b["image_aug"].shape == b["image"].shape \
                and np.array_equal(b["image_aug"], b["image"])
            if "image_aug" in b and not alias_aug:
                arrays["image_aug"] = np.ascontiguousarray(b["image_aug"])
            if shm is None:
                layout, off = [], 0
                for k, v in arrays.items():
                    layout.append((k, v.shape, v.dtype.str, off))
                    off += (v.nbytes + 255) // 256 * 256
                slot_bytes = max(off, 256)
                shm = shared_memory.SharedMemory(create=True, size=slot_bytes * nslots)
                views = [{k: np.ndarray(sh, dtype=np.dtype(dt), buffer=shm.buf, offset=sl * slot_bytes + o) for k, sh, dt, o in layout}
                         for sl in range(nslots)]
                q.put(("layout", shm.name, slot_bytes, nslots, layout))
            slot = free_q.get()                       # blocks until the parent has handed a slot back
            if slot is None:                          # the parent stopped early (the normal end of a repeat=True stream):
                stopped = True                        # it sends ONE token per worker -- do not wait for a second one below
                break
            for k, v in arrays.items():
                np.copyto(views[slot][k], v)
            q.put(("batch", slot, others, alias_aug))
        q.put(None)
        while not stopped and free_q.get() is not None:      # keep the segment alive until the parent says it is done with it
            pass
    except BaseException as e:               # surfaced in the parent as (type name, traceback text): always picklable
        import traceback
        q.put((_ERR, type(e).__name__, traceback.format_exc()))
    finally:
        if shm is not None:
            views = None
            try:
                shm.close()
                shm.unlink()
            except Exception:
                pass


def _batches_mp(ds_kw, files, seed, shuffle: bool, shuffle_buffer: int, repeat: bool, training: bool, batch: int, procs: int,
                threads: int = 2, nslots: int = 4):
    """Batches from ``procs`` worker PROCESSES (the thread pool of _examples stops scaling at ~8 threads: the Example parse
    and the NumPy glue hold the GIL).  Worker w owns shards ``files[w::procs]`` with its own shuffle buffer and random
    streams ([*seed, w]) and assembles whole batches; the parent takes one batch from each live worker in turn -- a
    deterministic interleave of independent streams (tf.data ``interleave`` over shards, base_dataset.py:60-72).  Batches
    arrive in a per-worker ring of shared-memory slots (``SlotBatch``): the consumer copies them out and releases the slot."""
    import multiprocessing as mp
    from multiprocessing import shared_memory
    ctx = mp.get_context("spawn")            # never fork a process that may already hold a HIP context
    procs = max(1, min(procs, len(files)))
    qs, fqs, ps = [], [], []
    for w in range(procs):
        q, fq = ctx.Queue(), ctx.Queue()
        for sl in range(nslots):
            fq.put(sl)
        pr = ctx.Process(target=_mp_worker, daemon=True,
                         args=(q, fq, ds_kw, files[w::procs], [*np.atleast_1d(seed).tolist(), w], shuffle, shuffle_buffer,
                               repeat, training, threads, batch, nslots))
        pr.start()
        qs.append(q)
        fqs.append(fq)
        ps.append(pr)
    shms, views = [None] * procs, [None] * procs
    live = list(range(procs))

    def get(w):
        while True:                   # a worker that died hard (OOM kill, a crash in the C decoder) never posts again
            try:
                return qs[w].get(timeout=2.0)
            except queue.Empty:
                if not ps[w].is_alive():
                    try:
                        return qs[w].get(timeout=0.5)      # it may have posted its last item just before exiting
                    except queue.Empty:
                        raise RuntimeError(f"input-pipeline worker {w} died (exit code {ps[w].exitcode}) without "
                                           f"finishing its shards {files[w::procs][:3]}...") from None
    try:
        while live:
            for w in list(live):
                item = get(w)
                if item is not None and item[0] == "layout":
                    _, name, slot_bytes, ns, layout = item
                    shms[w] = shared_memory.SharedMemory(name=name)
                    views[w] = [{k: np.ndarray(sh, dtype=np.dtype(dt), buffer=shms[w].buf, offset=sl * slot_bytes + o)
                                 for k, sh, dt, o in layout} for sl in range(ns)]
                    item = get(w)
                if item is None:
                    live.remove(w)
                elif item[0] == _ERR:
                    raise RuntimeError(f"input-pipeline worker {w} failed with {item[1]}:\n{item[2]}")
                else:
                    _, slot, others, alias_aug = item
                    out = SlotBatch(views[w][slot])
                    out.update(others)
                    if alias_aug:
                        out["image_aug"] = out["image"]
                    out._release = (lambda fq=fqs[w], sl=slot: fq.put(sl))
                    yield out
    finally:
        for w, pr in enumerate(ps):
            try:
                fqs[w].put(None)
            except Exception:
                pass
        views = None
        deadline = time.monotonic() + 2.0            # ONE shared deadline: the workers exit in parallel, not 1 s each in turn
        for pr in ps:
            pr.join(timeout=max(0.0, deadline - time.monotonic()))
        for pr in ps:
            if pr.is_alive():
                pr.terminate()
        for sh in shms:
            if sh is not None:
                try:
                    sh.close()
                except Exception:
                    pass
                try:                                 # backstop: a worker that was terminated never ran its own unlink
                    sh.unlink()
                except Exception:
                    pass


def _batches(examples, batch: int, drop_remainder: bool = True) -> Iterator[Dict[str, np.ndarray]]:
    cur = []
    for ex in examples:
        cur.append(ex)
        if len(cur) == batch:
            yield {k: np.stack([e[k] for e in cur]) if not isinstance(cur[0][k], (bytes, str)) else [e[k] for e in cur]
                   for k in cur[0]}
            cur = []
    if cur and not drop_remainder:
        yield {k: np.stack([e[k] for e in cur]) if not isinstance(cur[0][k], (bytes, str)) else [e[k] for e in cur]
               for k in cur[0]}


class Prefetcher:
    """Runs a batch iterator on a background thread, ``depth`` batches ahead; with ``device`` it also uploads
    every array through a pinned staging tensor on a side copy stream and yields device tensors."""

    def __init__(self, it, depth: int = 2, device=None):
        self._q = queue.Queue(maxsize=max(1, depth))
        self._device = device
        self._done = object()
        self._err = None
        self._thread = threading.Thread(target=self._run, args=(it,), daemon=True)
        self._thread.start()

    def _upload(self, batch):
        import torch
        stream = torch.cuda.Stream(device=self._device) if not hasattr(self, "_stream") else self._stream
        self._stream = stream
        out, seen = {}, {}
        with torch.cuda.stream(stream):
            for k, v in batch.items():
                if isinstance(v, np.ndarray):
                    if id(v) not in seen:            # aliased fields (image_aug is image) are uploaded once
                        seen[id(v)] = torch.from_numpy(v).pin_memory().to(self._device, non_blocking=True)   # pin_memory copies
                    out[k] = seen[id(v)]
                else:
                    out[k] = v
        if isinstance(batch, SlotBatch):
            batch.release()                          # the pinned copies are complete: the worker may refill the slot
        ev = torch.cuda.Event()
        ev.record(stream)
        return out, ev

    def _run(self, it):
        try:
            for b in it:
                if self._device is None and isinstance(b, SlotBatch):
                    own, seen = {}, {}
                    for k, v in b.items():           # copy out of the shared-memory slot, keeping aliases aliased
                        if isinstance(v, np.ndarray):
                            if id(v) not in seen:
                                seen[id(v)] = np.array(v)
                            own[k] = seen[id(v)]
                        else:
                            own[k] = v
                    b.release()
                    b = own
                self._q.put(self._upload(b) if self._device is not None else (b, None))
        except BaseException as e:          # surfaced on the consumer side
            self._err = e
        self._q.put(self._done)

    def __iter__(self):
        return self

    def __next__(self):
        item = self._q.get()
        if item is self._done:
            if self._err is not None:
                raise self._err
            raise StopIteration
        batch, ev = item
        if ev is not None:
            import torch
            cur = torch.cuda.current_stream()
            cur.wait_event(ev)                                   # the step's stream sees completed uploads
            for t in batch.values():
                # the tensors were allocated on the copy stream: tell the caching allocator the consumer stream uses
                # them, or a dropped batch's blocks could be re-filled by the next upload under still-queued kernels
                if isinstance(t, torch.Tensor):
                    t.record_stream(cur)
        return batch


def cpu_budget() -> float:
    """cores this process may use: the cgroup CPU quota (v2 ``cpu.max``, v1 ``cpu.cfs_quota_us``) or the affinity mask, whichever
    is smaller -- a 256-core host behind a 16-core quota decodes at the quota's rate, and more runnable decode threads than that
    only add throttling (round 5, 16-core quota: 16 x 1 threads 5.2 k examples/s, 16 x 2 threads 4.5 k)."""
    cores = float(len(os.sched_getaffinity(0))) if hasattr(os, "sched_getaffinity") else float(os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = min(cores, float(q) / float(per))
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                cores = min(cores, q / per)
        except (OSError, ValueError):
            pass
    return max(1.0, cores)


def decode_layout(procs: int, threads: int, budget: float = None, shards: int = None):
    """(processes, threads per process) actually started.  ``procs`` < 0: one single-threaded process per core of the budget
    (the fastest layout measured).  Otherwise the process count is kept (it defines the interleave of the shards, i.e. the
    data order) and the threads per process are cut so that the decode threads stay within 1.5 x the budget -- beyond that the
    rate FALLS (measured; a quota throttles every thread of the group once it is spent).  ``shards``: this rank's shard count --
    a process owns whole shards, so the count is clamped to it BEFORE the thread cap is computed.  NOTE ``procs`` < 0 makes the
    process count, hence the shard interleave and the order of the training data, a function of the host's CPU budget: use a
    fixed positive ``num_decode_procs`` for host-to-host reproducibility of a seed (create_datasets logs the resolved layout)."""
    budget = cpu_budget() if budget is None else float(budget)
    if procs < 0:
        n = max(1, int(budget))
        return (min(n, shards) if shards else n), 1
    if procs > 0 and shards:
        procs = min(procs, shards)
    if procs == 0:
        return 0, max(1, threads)
    threads = max(1, threads)
    cap = max(1, int(1.5 * budget / procs))
    return procs, min(threads, cap)


def create_datasets(config, data_rng: int = 0, rank: int = 0, world: int = 1, device=None, prefetch: int = 2,
                    workers: int = None, procs: int = None):
    """-> (train_iter, eval_iter, num_train_examples) -- reference input_pipeline.py:30-110.  ``rank`` is folded into
    every random stream (shuffle order, z, flips, caption choice), as the reference folds ``jax.host_id()`` into
    ``data_rng`` (train_utils.py:333); ``workers`` = decode threads (default ``config.num_decode_workers`` or 4) per
    process; ``procs`` (default ``config.num_decode_procs`` or 0) > 0 decodes the TRAINING stream in that many worker
    processes, each owning a subset of this rank's shards (measured on the MI355X box's host: one process saturates at
    ~0.9 k examples/s however many threads it has; the C1 step consumes 2.7 k/s per GPU)."""
    if config.batch_size % world != 0:
        raise ValueError(f"Batch size ({config.batch_size}) must be divisible by the number of devices ({world}).")
    per_device = config.batch_size // world
    per_device_train = per_device * config.d_step_per_g_step                # input_pipeline.py:46-47
    dtype = np.float32                                                       # the step casts to bf16 on the device
    if config.get("dataset", "mscoco") != "mscoco":
        raise NotImplementedError
    ds = COCODataset(image_size=config.image_size, z_dim=config.z_dim, data_dtype=dtype,
                     data_dir=config.get("data_dir", "data/"), coco_version=config.get("coco_version", "2014"),
                     return_text=config.get("return_text", False), return_filename=config.get("return_filename", False))
    seed = int(data_rng)
    if workers is None:
        workers = int(config.get("num_decode_workers", 4))

    def shard(split):
        files = ds.files(split)
        if len(files) < world:
            raise ValueError(f"{split}: {len(files)} TFRecord shard(s) for {world} ranks -- every rank would read the same "
                             f"records; re-shard the dataset or lower the number of processes")
        return files[rank::world]
    if procs is None:
        procs = int(config.get("num_decode_procs", 0))
    ds_kw = dict(image_size=config.image_size, z_dim=config.z_dim, data_dtype=dtype, data_dir=config.get("data_dir", "data/"),
                 coco_version=config.get("coco_version", "2014"), return_text=config.get("return_text", False),
                 return_filename=config.get("return_filename", False))
    sb = int(config.get("shuffle_buffer_size", 1000))
    # (ranks of one host share its cores: torchrun exports LOCAL_WORLD_SIZE)
    requested = procs
    procs, workers_mp = decode_layout(procs, workers, cpu_budget() / max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1"))),
                                      shards=len(shard("train")))
    if procs > 0:
        workers = workers_mp
        if requested < 0:                   # the data order follows the process count: say which one this host resolved to
            import logging
            logging.getLogger(__name__).warning("num_decode_procs=%d resolved to %d decode processes x %d thread(s) on this host: the "
                                                "training-data order of a seed depends on that count", requested, procs, workers)
        train = _batches_mp(ds_kw, shard("train"), [seed, 0, rank], config.get("train_shuffle", True), sb, True, True,
                            per_device_train, procs, max(1, workers))
    else:
        train = _batches(_examples(ds, shard("train"), [seed, 0, rank], config.get("train_shuffle", True), sb, True, True,
                                   workers), per_device_train)
    evalb = _batches(_examples(ds, shard("val"), [seed, 1, rank], True, sb, True, False, workers),
                     max(1, config.get("eval_batch_size", per_device) // world))
    return (Prefetcher(train, prefetch, device), Prefetcher(evalb, prefetch, device), ds.num_examples["train"])

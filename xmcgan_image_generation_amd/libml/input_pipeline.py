"""Training / evaluation input pipeline (reference ``xmcgan/libml/input_pipeline.py:30-110``).

``create_datasets(config, data_rng)`` -> ``(train_iter, eval_iter, num_train_examples)``: each iterator yields the
per-device batch dict of the step -- leading dim ``per_device_batch * d_step_per_g_step`` for training
(input_pipeline.py:46-47) -- built by ``COCODataset`` from the reference's sharded TFRecords: shard ``rank`` of
``world`` reads files ``rank::world`` (one process per GPU), a shuffle buffer, endless repetition for training.
A background thread decodes ahead of the consumer (``prefetch`` batches) and, on a GPU, hands the batch over as
device tensors through pinned host buffers on a copy stream, so the step's inputs are resident in HBM when it starts.
"""
from __future__ import annotations

import queue
import threading
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


def _mp_worker(q, ds_kw, files, seed, shuffle, shuffle_buffer, repeat, training, threads, batch):
    """child process of _batches_mp: decode its share of the shards, assemble whole batches, ship them as torch tensors
    in shared memory (the parent maps them, no copy through the pipe)"""
    try:
        import torch
        torch.set_num_threads(1)
        ds = COCODataset(**ds_kw)
        for b in _batches(_examples(ds, files, seed, shuffle, shuffle_buffer, repeat, training, threads), batch):
            q.put({k: (torch.from_numpy(np.ascontiguousarray(v)).share_memory_() if isinstance(v, np.ndarray) else v)
                   for k, v in b.items()})
        q.put(None)
    except BaseException as e:               # surfaced in the parent as (type name, traceback text): always picklable
        import traceback
        q.put(("__xmc_worker_error__", type(e).__name__, traceback.format_exc()))


def _batches_mp(ds_kw, files, seed, shuffle: bool, shuffle_buffer: int, repeat: bool, training: bool, batch: int, procs: int,
                threads: int = 2):
    """Batches from ``procs`` worker PROCESSES (the thread pool of _examples stops scaling at ~8 threads: the Example parse
    and the NumPy glue hold the GIL).  Worker w owns shards ``files[w::procs]`` with its own shuffle buffer and random
    streams ([*seed, w]) and assembles whole batches; the parent takes one batch from each live worker in turn -- a
    deterministic interleave of independent streams (tf.data ``interleave`` over shards, base_dataset.py:60-72).  Batches
    travel as shared-memory torch tensors (``torch.multiprocessing``), viewed as NumPy arrays on this side."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")            # never fork a process that may already hold a HIP context
    procs = max(1, min(procs, len(files)))
    qs, ps = [], []
    for w in range(procs):
        q = ctx.Queue(maxsize=3)
        pr = ctx.Process(target=_mp_worker, daemon=True,
                         args=(q, ds_kw, files[w::procs], [*np.atleast_1d(seed).tolist(), w], shuffle, shuffle_buffer,
                               repeat, training, threads, batch))
        pr.start()
        qs.append(q)
        ps.append(pr)
    live = list(range(procs))
    try:
        while live:
            for w in list(live):
                while True:                   # a worker that died hard (OOM kill, a crash in the C decoder) never posts again
                    try:
                        item = qs[w].get(timeout=2.0)
                        break
                    except queue.Empty:
                        if not ps[w].is_alive():
                            try:
                                item = qs[w].get(timeout=0.5)      # it may have posted its last item just before exiting
                                break
                            except queue.Empty:
                                raise RuntimeError(f"input-pipeline worker {w} died (exit code {ps[w].exitcode}) without "
                                                   f"finishing its shards {files[w::procs][:3]}...") from None
                if item is None:
                    live.remove(w)
                elif isinstance(item, tuple) and len(item) == 3 and item[0] == "__xmc_worker_error__":
                    raise RuntimeError(f"input-pipeline worker {w} failed with {item[1]}:\n{item[2]}")
                else:
                    yield {k: (v.numpy() if hasattr(v, "numpy") else v) for k, v in item.items()}
    finally:
        for pr in ps:
            pr.terminate()


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
        out = {}
        with torch.cuda.stream(stream):
            for k, v in batch.items():
                if isinstance(v, np.ndarray):
                    out[k] = torch.from_numpy(v).pin_memory().to(self._device, non_blocking=True)
                else:
                    out[k] = v
        ev = torch.cuda.Event()
        ev.record(stream)
        return out, ev

    def _run(self, it):
        try:
            for b in it:
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
    if procs > 0:
        train = _batches_mp(ds_kw, shard("train"), [seed, 0, rank], config.get("train_shuffle", True), sb, True, True,
                            per_device_train, procs, max(1, workers))
    else:
        train = _batches(_examples(ds, shard("train"), [seed, 0, rank], config.get("train_shuffle", True), sb, True, True,
                                   workers), per_device_train)
    evalb = _batches(_examples(ds, shard("val"), [seed, 1, rank], True, sb, True, False, workers),
                     max(1, config.get("eval_batch_size", per_device) // world))
    return (Prefetcher(train, prefetch, device), Prefetcher(evalb, prefetch, device), ds.num_examples["train"])

"""Input pipeline: TFRecord framing, tf.train.Example wire format, PNG decode, bilinear resize,
COCODataset.preprocess and the batch iterator -- pinned against independent implementations available in this image
(the protobuf runtime, PIL, torch's bilinear interpolation) and known answers.  CPU only."""
import io
import os
import struct

import numpy as np
import pytest
import torch

from xmcgan_image_generation_amd.configs import coco_xmc
from xmcgan_image_generation_amd.libml import _io, augmentation, coco_dataset, input_pipeline, png, tfrecord


def test_crc32c_known_answers():
    assert _io.crc32c(b"123456789") == 0xE3069283                    # CRC-32C check value (RFC 3720 B.4)
    assert _io.crc32c(b"") == 0
    assert _io.crc32c(bytes(32)) == 0x8A9136AA                       # RFC 3720 B.4: 32 bytes of zeros
    assert _io.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43              # RFC 3720 B.4: 32 bytes of 0xFF
    assert _io.crc32c(bytes(range(32))) == 0x46DD794E                # RFC 3720 B.4: 0..31
    c = _io.crc32c(b"abc")
    assert _io.masked_crc32c(b"abc") == (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def test_crc32c_hardware_path_equals_table_path():
    rng = np.random.default_rng(0)
    for n in (0, 1, 7, 8, 9, 15, 16, 17, 1000, 65537):
        d = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert _io.crc32c(d) == _io.crc32c_table(d)


def test_crc32_ieee_folded_equals_zlib():
    """the PNG chunk checksum by carry-less multiplication (csrc_host/xmc_io.c) against zlib.crc32 on every short length (the
    64-byte fold, the 16-byte blocks and the byte tail all take part) and on long buffers"""
    import zlib
    rng = np.random.default_rng(0)
    for n in list(range(0, 300)) + [1000, 4095, 4096, 4097, 65536, 590371, 1 << 20]:
        d = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert _io.crc32_ieee(d) == zlib.crc32(d), n


def test_tfrecord_round_trip_and_corruption(tmp_path):
    recs = [b"", b"x", os.urandom(1000), os.urandom(70000)]
    path = str(tmp_path / "a.tfrecord")
    tfrecord.write_records(path, recs)
    assert list(tfrecord.read_records(path)) == recs
    raw = bytearray(open(path, "rb").read())
    raw[-100] ^= 1                                                    # flip one payload bit of the last record
    open(path, "wb").write(raw)
    with pytest.raises(IOError):
        list(tfrecord.read_records(path))
    assert len(list(tfrecord.read_records(path, verify_crc=False))) == 4


def _example_proto_classes():
    """tf.train.Example message classes built at run time from the published tensorflow/core/example/*.proto schema
    (field numbers only) with the protobuf runtime -- an implementation independent of libml/tfrecord.py."""
    pb = pytest.importorskip("google.protobuf")
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    fd = descriptor_pb2.FileDescriptorProto(name="xmc_test_example.proto", package="xmctest", syntax="proto3")
    T = descriptor_pb2.FieldDescriptorProto

    def msg(name, fields, nested=()):
        m = fd.message_type.add(name=name)
        for fname, num, typ, label, tname in fields:
            f = m.field.add(name=fname, number=num, type=typ, label=label)
            if tname:
                f.type_name = tname
        return m
    msg("BytesList", [("value", 1, T.TYPE_BYTES, T.LABEL_REPEATED, None)])
    msg("FloatList", [("value", 1, T.TYPE_FLOAT, T.LABEL_REPEATED, None)])
    msg("Int64List", [("value", 1, T.TYPE_INT64, T.LABEL_REPEATED, None)])
    feat = msg("Feature", [("bytes_list", 1, T.TYPE_MESSAGE, T.LABEL_OPTIONAL, ".xmctest.BytesList"),
                           ("float_list", 2, T.TYPE_MESSAGE, T.LABEL_OPTIONAL, ".xmctest.FloatList"),
                           ("int64_list", 3, T.TYPE_MESSAGE, T.LABEL_OPTIONAL, ".xmctest.Int64List")])
    feat.oneof_decl.add(name="kind")
    for f in feat.field:
        f.oneof_index = 0
    feats = msg("Features", [("feature", 1, T.TYPE_MESSAGE, T.LABEL_REPEATED, ".xmctest.Features.FeatureEntry")])
    entry = feats.nested_type.add(name="FeatureEntry")
    entry.field.add(name="key", number=1, type=T.TYPE_STRING, label=T.LABEL_OPTIONAL)
    entry.field.add(name="value", number=2, type=T.TYPE_MESSAGE, label=T.LABEL_OPTIONAL, type_name=".xmctest.Feature")
    entry.options.map_entry = True
    msg("Example", [("features", 1, T.TYPE_MESSAGE, T.LABEL_OPTIONAL, ".xmctest.Features")])
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    return message_factory.GetMessageClass(pool.FindMessageTypeByName("xmctest.Example"))


def test_example_wire_format_against_protobuf_runtime():
    Example = _example_proto_classes()
    rng = np.random.default_rng(0)
    emb = rng.standard_normal(5 * 17 * 8).astype(np.float32)
    feats = {"image": [b"\x89PNG-bytes"], "image/filename": [b"COCO_train2014_000000000009.jpg"],
             "caption/text": [b"a cat", b"two dogs", "café".encode()],
             "caption/embedding": emb, "caption/max_len": np.array([4, 17, 9, -3, 2 ** 40], np.int64)}
    # protobuf writes, this module reads
    ex = Example()
    for k, v in feats.items():
        f = ex.features.feature[k]
        if isinstance(v, list):
            f.bytes_list.value.extend(v)
        elif v.dtype.kind == "f":
            f.float_list.value.extend(v.tolist())
        else:
            f.int64_list.value.extend(v.tolist())
    got = tfrecord.parse_example(ex.SerializeToString())
    assert set(got) == set(feats)
    for k, v in feats.items():
        if isinstance(v, list):
            assert got[k] == v
        else:
            assert np.array_equal(got[k], v) and got[k].dtype == v.dtype
    # this module writes, protobuf reads
    ex2 = Example()
    ex2.ParseFromString(tfrecord.serialize_example(feats))
    assert list(ex2.features.feature["caption/text"].bytes_list.value) == feats["caption/text"]
    assert np.array_equal(np.array(ex2.features.feature["caption/embedding"].float_list.value, np.float32), emb)
    assert list(ex2.features.feature["caption/max_len"].int64_list.value) == feats["caption/max_len"].tolist()


def test_png_decode_against_pil_all_filters_and_colour_types():
    PIL = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, size=(37, 53, 3), dtype=np.uint8)
    img[:, :20] = (np.arange(20)[None, :, None] * 9 + np.arange(37)[:, None, None]).astype(np.uint8)   # smooth part: filters matter
    for mode in ("RGB", "RGBA", "L", "LA", "P"):
        buf = io.BytesIO()
        PIL.fromarray(img).convert(mode).save(buf, format="PNG", optimize=True)
        want = np.asarray(PIL.open(io.BytesIO(buf.getvalue())).convert("RGB"))
        assert np.array_equal(png.decode_rgb(buf.getvalue()), want), mode
    # our own encoder, each of the five scanline filters; PIL must read it back bit-exactly too
    for ft in range(5):
        data = png.encode_rgb(img, filter_types=[ft] * img.shape[0])
        assert np.array_equal(png.decode_rgb(data), img), ft
        assert np.array_equal(np.asarray(PIL.open(io.BytesIO(data)).convert("RGB")), img), ft
    data = png.encode_rgb(img, filter_types=np.arange(img.shape[0]) % 5)
    assert np.array_equal(png.decode_rgb(data), img)
    with pytest.raises(ValueError):
        png.decode_rgb(data[:40] + bytes([data[40] ^ 1]) + data[41:])          # chunk crc


@pytest.mark.parametrize("shape", [(480, 640), (97, 61), (128, 128), (64, 300)])
def test_resize_matches_half_pixel_bilinear(shape):
    """tf.image.resize(method="bilinear") of TF2 = half-pixel centres, no anti-aliasing = torch's
    F.interpolate(mode="bilinear", align_corners=False, antialias=False)."""
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, size=(*shape, 3), dtype=np.uint8)
    got = _io.resize_bilinear_rgb(img, 128, flip=False)
    x = torch.from_numpy(img).permute(2, 0, 1)[None].float() / 255.0
    want = torch.nn.functional.interpolate(x, size=(128, 128), mode="bilinear", align_corners=False)[0].permute(1, 2, 0)
    assert got.shape == (128, 128, 3) and got.dtype == np.float32
    assert np.abs(got - want.numpy()).max() < 2e-6
    assert np.array_equal(_io.resize_bilinear_rgb(img, 128, flip=True), got[:, ::-1])
    assert got.min() >= 0.0 and got.max() <= 1.0


def _write_shards(tmp_path, n=7, split="train", seed=3):
    rng = np.random.default_rng(seed)
    exs = []
    for i in range(n):
        h, w = int(rng.integers(40, 90)), int(rng.integers(40, 90))
        img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        emb = rng.standard_normal((5, 17, 768)).astype(np.float32)
        ml = rng.integers(3, 18, size=5).astype(np.int64)
        exs.append(dict(img=img, emb=emb, ml=ml, name=f"img{i}.jpg".encode()))
    name = {"train": "train", "val": "validation"}[split]
    for shard in range(2):
        recs = [tfrecord.serialize_example({"image": [png.encode_rgb(e["img"], np.arange(e["img"].shape[0]) % 5)],
                                            "image/filename": [e["name"]],
                                            "caption/text": [f"caption {k}".encode() for k in range(5)],
                                            "caption/embedding": e["emb"].reshape(-1), "caption/max_len": e["ml"]})
                for e in exs[shard::2]]
        tfrecord.write_records(str(tmp_path / f"coco2014_{name}.tfrecord-{shard}-of-2"), recs)
    return exs


def test_coco_dataset_preprocess_contract(tmp_path):
    exs = _write_shards(tmp_path)
    ds = coco_dataset.COCODataset(image_size=128, z_dim=8, data_dir=str(tmp_path) + "/", return_filename=True)
    files = ds.files("train")
    assert len(files) == 2
    rec = next(tfrecord.read_records(files[0]))
    f = ds.parse_example(rec)
    assert np.array_equal(f["image"], exs[0]["img"]) and f["caption/embedding"].shape == (5, 17, 768)
    assert np.array_equal(f["caption/max_len"], exs[0]["ml"]) and f["caption/text"][2] == b"caption 2"
    a, b = ds.preprocess(f, 11), ds.preprocess(f, 11)
    c = ds.preprocess(f, 12)
    for k in ("image", "image_aug", "embedding", "max_len", "sentence_embedding", "z"):
        assert np.array_equal(a[k], b[k]), k                                     # same rng -> same example
    assert not np.array_equal(a["z"], c["z"])
    assert a["image"].shape == (128, 128, 3) and a["image_aug"].shape == (128, 128, 3) and a["z"].shape == (8,)
    assert a["embedding"].shape == (17, 768) and a["max_len"].shape == (1,) and a["sentence_embedding"].shape == (768,)
    assert a["image"].min() >= 0.0 and a["image"].max() <= 1.0 and a["filename"] == b"img0.jpg"
    # the selected caption: embedding row idx, max_len[idx], sentence = sum over ALL 17 rows / max_len (coco_dataset.py:142)
    idx = [i for i in range(5) if np.array_equal(a["embedding"], exs[0]["emb"][i])]
    assert len(idx) == 1 and float(a["max_len"][0]) == float(exs[0]["ml"][idx[0]])
    assert np.allclose(a["sentence_embedding"], exs[0]["emb"][idx[0]].sum(0) / exs[0]["ml"][idx[0]], rtol=1e-6)
    # the image is the resized (possibly flipped) original
    base = _io.resize_bilinear_rgb(exs[0]["img"], 128)
    assert np.array_equal(a["image"], base) or np.array_equal(a["image"], base[:, ::-1])
    # return_text: the SHORTEST caption (coco_dataset.py:151-153)
    ds_t = coco_dataset.COCODataset(image_size=128, z_dim=8, data_dir=str(tmp_path) + "/", return_text=True)
    t = ds_t.preprocess(ds_t.parse_example(rec), 5)
    assert float(t["max_len"][0]) == float(exs[0]["ml"].min()) and t["text"].startswith(b"caption")
    with pytest.raises(ValueError):
        ds.get_file_patterns("test")


def test_augmentation_shape_and_determinism():
    """mirrors the reference's augmentation_test.py: output shape preserved, same seed -> same result"""
    rng = np.random.default_rng(4)
    x = rng.random((2, 32, 32, 3)).astype(np.float32)
    for fn in (augmentation.augment_shift, augmentation.augment_zoom_crop, augmentation.augment):
        a, b, c = fn(x, seed=7), fn(x, seed=7), fn(x, seed=8)
        assert a.shape == x.shape and np.array_equal(a, b) and not np.array_equal(a, c), fn.__name__
    with pytest.raises(NotImplementedError):
        augmentation.augment(x, method="rotate", seed=1)
    with pytest.raises(NotImplementedError):
        augmentation.augment_zoom_crop(x, resize_method="bicubic", seed=1)


def test_create_datasets_feeds_the_training_step(tmp_path):
    """pipeline -> batch dict of leading dim per_device_batch * d_step_per_g_step -> one train_step (CPU mock ops)"""
    from tests.cpu_ops import CpuOps
    from xmcgan_image_generation_amd import train_utils, xmc_gan
    from xmcgan_image_generation_amd.nets import xmc_net
    _write_shards(tmp_path, n=9, split="train")
    _write_shards(tmp_path, n=4, split="val", seed=5)
    cfg = coco_xmc.get_test_config()
    cfg.batch_size = 2
    cfg.update(data_dir=str(tmp_path) + "/", coco_version="2014", shuffle_buffer_size=4, train_shuffle=True,
               eval_batch_size=2, dataset="mscoco")
    tr, ev, n = input_pipeline.create_datasets(cfg, data_rng=3)
    assert n == 82_783
    b1, b2 = next(tr), next(tr)
    assert b1["image"].shape == (4, 128, 128, 3) and b1["embedding"].shape == (4, 17, 768) and b1["z"].shape == (4, cfg.z_dim)
    assert b1["max_len"].shape == (4, 1) and b1["sentence_embedding"].shape == (4, 768)
    assert not np.array_equal(b1["image"], b2["image"])
    tr2, _, _ = input_pipeline.create_datasets(cfg, data_rng=3)
    assert np.array_equal(next(tr2)["image"], b1["image"])                        # same seed -> same stream
    assert next(ev)["image"].shape == (2, 128, 128, 3)
    xmc_net.set_ops_factory(lambda dtype: CpuOps(dtype))
    try:
        gen, disc, state = train_utils.create_train_state(cfg, 0)
        tb = {k: torch.as_tensor(v) for k, v in b1.items()}
        state, metrics = train_utils.train_step(0, state, tb, xmc_gan, gen, disc, cfg, {})
        assert all(np.isfinite(float(v)) for v in metrics.values()) and state.step == 1
    finally:
        xmc_net.set_ops_factory(None)


def test_create_datasets_replicas_and_workers(tmp_path):
    """rank is folded into every random stream (reference train_utils.py:333 folds jax.host_id() into data_rng):
    two replicas draw different z; fewer shards than ranks raises instead of silently duplicating data; the decode
    thread pool (base_dataset.py:69-72 maps with num_parallel_calls=AUTOTUNE) preserves the single-thread stream."""
    _write_shards(tmp_path, n=9, split="train")
    _write_shards(tmp_path, n=4, split="val", seed=5)
    cfg = coco_xmc.get_test_config()
    cfg.batch_size = 4
    cfg.update(data_dir=str(tmp_path) + "/", coco_version="2014", shuffle_buffer_size=4, train_shuffle=True,
               eval_batch_size=2, dataset="mscoco")
    r0, _, _ = input_pipeline.create_datasets(cfg, data_rng=3, rank=0, world=2)
    r1, _, _ = input_pipeline.create_datasets(cfg, data_rng=3, rank=1, world=2)
    b0, b1 = next(r0), next(r1)
    assert b0["z"].shape == (4, cfg.z_dim)                      # per-device 2 x d_step_per_g_step 2
    for i in range(4):
        for j in range(4):
            assert not np.array_equal(b0["z"][i], b1["z"][j]), (i, j)
    assert not np.array_equal(b0["image"], b1["image"])         # shard rank::world
    with pytest.raises(ValueError, match="shard"):
        input_pipeline.create_datasets(cfg, data_rng=3, rank=0, world=4)
    cfg.batch_size = 2
    one, _, _ = input_pipeline.create_datasets(cfg, data_rng=5, workers=1)
    four, _, _ = input_pipeline.create_datasets(cfg, data_rng=5, workers=4)
    for _ in range(3):
        a, b = next(one), next(four)
        for k in ("image", "z", "embedding", "max_len", "sentence_embedding"):
            assert np.array_equal(a[k], b[k]), k


def test_create_datasets_worker_processes(tmp_path):
    """the multi-process decode path: deterministic (same seed -> same stream), every example a valid preprocess output"""
    _write_shards(tmp_path, n=8, split="train")
    _write_shards(tmp_path, n=4, split="val", seed=5)
    cfg = coco_xmc.get_test_config()
    cfg.batch_size = 2
    cfg.update(data_dir=str(tmp_path) + "/", coco_version="2014", shuffle_buffer_size=2, train_shuffle=True,
               eval_batch_size=2, dataset="mscoco")
    a, _, _ = input_pipeline.create_datasets(cfg, data_rng=7, workers=1, procs=2)
    b, _, _ = input_pipeline.create_datasets(cfg, data_rng=7, workers=2, procs=2)
    for _ in range(2):
        x, y = next(a), next(b)
        for k in ("image", "z", "embedding", "max_len", "sentence_embedding"):
            assert np.array_equal(x[k], y[k]), k
        assert x["image"].shape == (4, 128, 128, 3) and 0.0 <= x["image"].min() and x["image"].max() <= 1.0


def test_worker_processes_shut_down_cleanly_on_early_stop(tmp_path):
    """the normal end of a repeat=True training stream: the consumer closes the generator while every worker is blocked on its
    free-slot queue.  One stop token per worker must be enough (round-4 advisor finding: a worker waited for a second one and
    was terminated after a 1 s join each, its shared-memory ring left in /dev/shm): the workers exit by themselves well inside
    the join deadline and unlink their segments."""
    import glob
    import time
    _write_shards(tmp_path, n=8, split="train")
    ds_kw = dict(image_size=128, z_dim=8, data_dir=str(tmp_path) + "/", return_text=False, return_filename=False)
    files = sorted(glob.glob(str(tmp_path) + "/*train*"))
    assert len(files) >= 2
    before = set(glob.glob("/dev/shm/psm_*"))
    gen = input_pipeline._batches_mp(ds_kw, files, [3], True, 2, True, True, 2, procs=2, threads=1, nslots=2)
    first = next(gen)
    assert first["image"].shape[0] == 2
    time.sleep(0.5)                                  # both workers have filled their rings and block on free_q.get()
    created = set(glob.glob("/dev/shm/psm_*")) - before
    assert created, "the workers' shared-memory rings exist while the stream is live"
    t0 = time.monotonic()
    gen.close()                                      # runs _batches_mp's finally block
    took = time.monotonic() - t0
    assert took < 1.5, f"worker shutdown took {took:.2f} s: the stop handshake hung until the join deadline"
    assert not (set(glob.glob("/dev/shm/psm_*")) & created), "a worker's shared-memory ring outlived the stream"


# ------------------------------------------------------------------ round 4: csrc_host/xmc_inflate.c + the SSE2 Paeth rows
def _zlib_streams():
    import zlib
    rng = np.random.default_rng(0)
    payloads = [b"", b"a", b"abc" * 1000, bytes(70000), rng.integers(0, 256, 70000, dtype=np.uint8).tobytes(),
                rng.integers(0, 4, 150000, dtype=np.uint8).tobytes(),
                (np.cumsum(rng.integers(-2, 3, 200000)) % 256).astype(np.uint8).tobytes(),
                b"".join(bytes([i % 251]) * (i % 300) for i in range(1500))]
    payloads += [rng.integers(0, 16, n, dtype=np.uint8).tobytes() for n in (1, 2, 7, 8, 9, 257, 258, 259, 32768, 32769)]
    for d in payloads:
        for level in (0, 1, 6, 9):                   # stored / fast / default / best: every block type
            for strat in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE):
                for wbits in (15, 9):
                    co = zlib.compressobj(level, zlib.DEFLATED, wbits, 9, strat)
                    yield d, co.compress(d) + co.flush()


def test_inflate_matches_zlib_on_every_block_type():
    """The one-shot DEFLATE decoder of the PNG path against zlib's own output: stored, fixed and dynamic blocks, runs
    (distance 1), short distances (2 .. 7: the word copy that advances by the distance), window sizes, empty input, streams cut
    by full flushes."""
    import zlib
    n = 0
    for d, c in _zlib_streams():
        assert _io.inflate_zlib(c, len(d)).tobytes() == d
        n += 1
    assert n > 500
    rng = np.random.default_rng(1)
    d = (np.cumsum(rng.integers(-3, 4, 300000)) % 256).astype(np.uint8).tobytes()
    co, c = zlib.compressobj(6), b""
    for i in range(0, len(d), 7777):
        c += co.compress(d[i:i + 7777])
        if (i // 7777) % 3 == 0:
            c += co.flush(zlib.Z_FULL_FLUSH)
    c += co.flush()
    assert _io.inflate_zlib(c, len(d)).tobytes() == d


def test_inflate_rejects_malformed_streams_without_leaving_its_buffers():
    """Truncations, bit flips, random bytes, wrong announced sizes: an error, never a crash; what is accepted after a single bit
    flip is what zlib decodes too (zlib additionally checks the Adler-32 trailer, which the PNG path covers with chunk CRCs)."""
    import zlib
    rng = np.random.default_rng(2)
    d = (np.cumsum(rng.integers(-3, 4, 120000)) % 256).astype(np.uint8).tobytes()
    c = zlib.compress(d, 6)
    for size in (len(d) - 1, len(d) + 1, 0):
        with pytest.raises(ValueError):
            _io.inflate_zlib(c, size)
    for _ in range(1500):
        cc = bytearray(c[:15000])
        k = int(rng.integers(0, 4))
        if k == 0:
            cc = cc[:int(rng.integers(0, len(cc)))]
        elif k == 1:
            for _ in range(int(rng.integers(1, 5))):
                cc[int(rng.integers(0, len(cc)))] ^= 1 << int(rng.integers(0, 8))
        elif k == 2:
            cc = bytearray(rng.integers(0, 256, int(rng.integers(0, 3000)), dtype=np.uint8).tobytes())
        else:
            cc[2:40] = rng.integers(0, 256, 38, dtype=np.uint8).tobytes()
        try:
            _io.inflate_zlib(bytes(cc), int(rng.integers(0, 200000)))
        except ValueError:
            pass
    for _ in range(800):
        cc = bytearray(c)
        cc[int(rng.integers(2, len(cc) - 4))] ^= 1 << int(rng.integers(0, 8))
        try:
            got = _io.inflate_zlib(bytes(cc), len(d)).tobytes()
        except ValueError:
            continue
        try:
            assert zlib.decompressobj().decompress(bytes(cc)) == got
        except zlib.error:
            pass


def _filter_rows(img, ft):
    """PNG filtering (spec 9.2) of an (h, w, c) uint8 image with the filter type of every row given"""
    h, w, ch = img.shape
    rows = img.reshape(h, w * ch).astype(np.int32)
    raw, prev = bytearray(), np.zeros(w * ch, np.int32)
    for y in range(h):
        cur = rows[y]
        left = np.concatenate([np.zeros(ch, np.int32), cur[:-ch]])
        ul = np.concatenate([np.zeros(ch, np.int32), prev[:-ch]])
        f = int(ft[y])
        if f == 0:
            o = cur
        elif f == 1:
            o = cur - left
        elif f == 2:
            o = cur - prev
        elif f == 3:
            o = cur - ((left + prev) >> 1)
        else:
            p = left + prev - ul
            pa, pb, pc = np.abs(p - left), np.abs(p - prev), np.abs(p - ul)
            o = cur - np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, ul))
        raw.append(f)
        raw += (o & 255).astype(np.uint8).tobytes()
        prev = cur
    return bytes(raw)


def test_unfilter_vector_paths_all_shapes():
    """The Paeth rows (single row; the two-row, the packed four-row SSE2 and the eight-row AVX2 wavefronts, with their scalar triangles
    and tails) and the scalar rows: every filter, 1-4 bytes per pixel, rows
    too short for the vector loop, predictor ties, odd / even row counts, Paeth runs broken by other filters."""
    rng = np.random.default_rng(3)
    for (h, w, ch) in ((7, 5, 3), (1, 1, 3), (3, 2, 3), (2, 4, 3), (2, 3, 3), (16, 17, 4), (5, 1, 4), (4, 4, 4), (9, 33, 1), (6, 7, 2),
                       (33, 129, 3), (64, 64, 4), (8, 16, 3), (9, 16, 3), (12, 15, 3), (11, 8, 4), (20, 40, 3), (15, 16, 4), (8, 7, 3)):
        for trial in range(5):
            img = rng.integers(0, 256, (h, w, ch), dtype=np.uint8)
            if trial == 1:
                img = (img // 64 * 64).astype(np.uint8)          # ties between the three predictors
            if trial == 3:
                img = np.zeros_like(img)
            ft = rng.integers(0, 5, h) if trial < 2 else (np.full(h, 4) if trial < 4 else rng.choice([4, 4, 4, 1, 2], h))
            out = _io.png_unfilter(_filter_rows(img, ft), h, w * ch, ch)
            assert (out.reshape(h, w, ch) == img).all(), (h, w, ch, trial)


def test_png_decode_fast_inflate_equals_zlib_path():
    rng = np.random.default_rng(4)
    yy, xx = np.mgrid[0:96, 0:130].astype(np.float32)
    img = np.stack([127 + 100 * np.sin(0.03 * xx + 0.02 * yy + k) for k in range(3)], -1) + rng.normal(0, 4, (96, 130, 3))
    img = np.clip(img, 0, 255).astype(np.uint8)
    for ft in (np.full(96, 4), rng.integers(0, 5, 96)):
        data = png.encode_rgb(img, ft)
        a, b = _io.png_decode(data)[0], _io.png_decode(data, use_zlib=True)[0]
        assert (a == img).all() and (b == img).all()
    bad = bytearray(png.encode_rgb(img, np.full(96, 4)))
    bad[len(bad) // 2] ^= 0x10
    with pytest.raises(ValueError):
        _io.png_decode(bytes(bad))                   # chunk CRC
    with pytest.raises(ValueError):
        _io.png_decode(bytes(bad), verify_crc=False)             # the inflater (or the size check) catches it


def test_png_header_claiming_a_huge_image_is_refused_before_any_allocation():
    """round-4 advisor finding: IHDR width / height are untrusted (up to 2^31 - 1 each); h * (w * ch + 1) overflowed and
    the Python wrapper allocated h * w * ch bytes from the header.  xmc_png_info now caps the decoded size at 1 GiB."""
    import struct
    import zlib
    good = bytearray(png.encode_rgb(np.zeros((4, 4, 3), np.uint8), np.zeros(4, np.int64)))
    for w, h in ((0x7fffffff, 0x7fffffff), (0x7fffffff, 1), (40000, 40000)):
        bomb = bytearray(good)
        bomb[16:24] = struct.pack(">II", w, h)
        bomb[29:33] = struct.pack(">I", zlib.crc32(bytes(bomb[12:29])) & 0xffffffff)     # a VALID header CRC
        with pytest.raises(ValueError, match="cap"):
            _io.png_decode(bytes(bomb))
    assert _io.png_decode(bytes(good))[0].shape == (4, 4, 3)


def test_decode_layout_stays_inside_the_cpu_budget():
    """More runnable decode threads than ~1.5 x the CPU quota LOWER the rate (measured on a 16-core quota: 16 x 1 threads 5.2 k
    examples/s, 16 x 2 threads 4.5 k): the process count is kept (it defines the data order), the threads per process are cut."""
    from xmcgan_image_generation_amd.libml import input_pipeline as ip
    assert ip.decode_layout(12, 2, 16) == (12, 2)
    assert ip.decode_layout(14, 2, 16) == (14, 1)
    assert ip.decode_layout(16, 2, 16) == (16, 1)
    assert ip.decode_layout(16, 4, 64) == (16, 4)
    assert ip.decode_layout(64, 2, 16) == (64, 1)
    assert ip.decode_layout(-1, 4, 16) == (16, 1)
    assert ip.decode_layout(0, 3, 16) == (0, 3)
    assert ip.cpu_budget() >= 1.0

"""COCO text-to-image dataset reader (reference ``xmcgan/libml/coco_dataset.py``).

``COCODataset.parse_example`` (:70-111) and ``preprocess`` (:127-167) on the host in NumPy + the C helpers of
``csrc_host/xmc_io.c``: TFRecord -> ``tf.train.Example`` -> PNG decode -> bilinear resize to ``image_size`` ->
random left-right flip -> clip -> caption selection -> the batch-dict contract the training step consumes
(``image, image_aug, embedding, max_len, sentence_embedding, z``).

Randomness: the reference uses TF stateless RNGs keyed per example; those streams cannot be reproduced without
TensorFlow, so a NumPy ``Generator`` per example (seeded from the pipeline seed and the example index) plays that
role -- deterministic for a given seed, not bit-equal to a TF run.
"""
from __future__ import annotations

import glob
from typing import Dict, Optional

import numpy as np

from . import _io, augmentation, png, tfrecord
from .. import synthetic as syn


class COCODataset:
    """Same constructor surface as the reference's ``COCODataset`` (base_dataset.py / coco_dataset.py:30-68)."""

    def __init__(self, image_size: int = 128, z_dim: int = 128, data_dtype=np.float32, data_dir: str = "data/",
                 coco_version: str = "2014", return_text: bool = False, return_filename: bool = False,
                 sentence_num: int = 5, max_text_length: int = syn.MAX_WORDS, embedding_dim: int = syn.EMB_DIM):
        self.image_size, self.z_dim, self.data_dtype = image_size, z_dim, data_dtype
        self.data_dir, self.coco_version = data_dir, coco_version
        self.return_text, self.return_filename = return_text, return_filename
        self.sentence_num = sentence_num
        self.embedding_shape = (sentence_num, max_text_length, embedding_dim)      # coco_dataset.py:62-63

    @property
    def num_examples(self):                                                       # coco_dataset.py:169-176
        return {"2017": {"train": 116_680, "val": 4_958}, "2014": {"train": 82_783, "val": 40_504},
                "ln": {"train": 134_272, "val": 8_573}}[self.coco_version]

    def get_file_patterns(self, split: Optional[str] = None, file_pattern: Optional[str] = None):
        if not file_pattern:
            if split not in ("train", "val"):
                raise ValueError(f"Expected split to be one of ['train', 'val'], got {split}")
            if split == "val":
                split = "validation"
            file_pattern = self.data_dir + f"*{self.coco_version}*{split}.tfrecord*"
        return file_pattern

    def files(self, split=None, file_pattern=None):
        out = sorted(glob.glob(self.get_file_patterns(split, file_pattern)))
        if not out:
            raise FileNotFoundError(self.get_file_patterns(split, file_pattern))
        return out

    def parse_example(self, example: bytes) -> Dict[str, object]:
        """coco_dataset.py:70-111: image decoded to float32 (H, W, 3) in [0, 1] lazily (uint8 kept for the resize)."""
        f = tfrecord.parse_example(example)
        emb = np.asarray(f["caption/embedding"], np.float32)
        if emb.size != int(np.prod(self.embedding_shape)):
            raise ValueError(f"caption/embedding has {emb.size} values, expected {self.embedding_shape}")
        return {"image": png.decode_rgb(f["image"][0]),                            # uint8 (H, W, 3)
                "image/filename": f["image/filename"][0] if f.get("image/filename") else b"",
                "caption/text": list(f.get("caption/text", [])),
                "caption/embedding": emb.reshape(self.embedding_shape),
                "caption/max_len": np.asarray(f["caption/max_len"], np.int64)}

    def preprocess(self, features: Dict[str, object], rng, training: bool = True) -> Dict[str, np.ndarray]:
        """coco_dataset.py:127-167.  ``rng``: np.random.Generator (or seed) of this example."""
        rng = rng if isinstance(rng, np.random.Generator) else np.random.default_rng(rng)
        rng_flip, rng_sent_idx, rng_z, rng_aug = (np.random.default_rng(s) for s in rng.integers(0, 2 ** 63 - 1, size=4))
        flip = bool(rng_flip.random() < 0.5)
        image = _io.resize_bilinear_rgb(features["image"], self.image_size, flip)   # resize, flip, clip to [0, 1]
        image_aug = augmentation.augment(image[None, ...], seed=rng_aug)[0]
        embedding = features["caption/embedding"]
        max_len = features["caption/max_len"].astype(np.float32)[:, None]           # (S, 1)
        sentence_feat = embedding.sum(axis=-2) / max_len                            # includes padded rows (:142)
        idx = int(rng_sent_idx.integers(0, self.sentence_num))
        if self.return_text:                                                        # shortest caption (:151-153)
            idx = int(np.argsort(-features["caption/max_len"], kind="stable")[-1])
        dt = self.data_dtype
        out = dict(image=image.astype(dt), image_aug=image_aug.astype(dt), embedding=embedding[idx].astype(dt),
                   max_len=max_len[idx].astype(dt), sentence_embedding=sentence_feat[idx].astype(dt))
        if self.return_text:
            out["text"] = features["caption/text"][idx]
        if self.return_filename:
            out["filename"] = features["image/filename"]
        out["z"] = rng_z.standard_normal((self.z_dim,)).astype(dt)
        return out

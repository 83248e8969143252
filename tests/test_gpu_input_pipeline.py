"""SURVEY.md 8(f) N4 on the HIP backend: TFRecord shards -> create_datasets(device="cuda") -> real train_steps.

The batch the step consumes is checked against COCODataset.preprocess on the same records (reference
coco_dataset.py:127-167 semantics, pinned on CPU in tests/test_input_pipeline.py), then fed to two train_steps."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_tfrecord_pipeline_feeds_the_hip_step(tmp_path):
    from tests.test_input_pipeline import _write_shards
    from xmcgan_image_generation_amd import train_utils, xmc_gan
    from xmcgan_image_generation_amd.configs import coco_xmc
    from xmcgan_image_generation_amd.libml import coco_dataset, input_pipeline, tfrecord
    _write_shards(tmp_path, n=12, split="train")
    _write_shards(tmp_path, n=4, split="val", seed=5)
    cfg = coco_xmc.get_test_config()
    cfg.batch_size = 2
    cfg.dtype = "bfloat16"
    cfg.update(data_dir=str(tmp_path) + "/", coco_version="2014", shuffle_buffer_size=4, train_shuffle=False,
               eval_batch_size=2, dataset="mscoco")
    train, _, _ = input_pipeline.create_datasets(cfg, data_rng=3, device="cuda", workers=4)
    # expected: the records in reading order (train_shuffle off), example i drawn from default_rng([seed, 0, rank, i])
    ds = coco_dataset.COCODataset(image_size=cfg.image_size, z_dim=cfg.z_dim, data_dir=cfg.data_dir)
    recs = [r for f in ds.files("train") for r in tfrecord.read_records(f)]
    want = [ds.preprocess(ds.parse_example(r), np.random.default_rng([3, 0, 0, i]), True) for i, r in enumerate(recs[:8])]
    gen, disc, state = train_utils.create_train_state(cfg, 0)
    for step in range(2):
        batch = next(train)
        assert all(batch[k].is_cuda for k in ("image", "embedding", "max_len", "sentence_embedding", "z"))
        assert batch["image"].shape == (4, cfg.image_size, cfg.image_size, 3)          # per-device 2 x d_step_per_g_step 2
        for k in ("image", "embedding", "max_len", "sentence_embedding", "z"):
            exp = np.stack([w[k] for w in want[4 * step:4 * step + 4]])
            assert np.array_equal(batch[k].cpu().numpy(), exp), (step, k)
        state, metrics = train_utils.train_step(step, state, batch, xmc_gan, gen, disc, cfg, {})
        del batch                                   # dropped while the step may still be queued (record_stream keeps it alive)
    torch.cuda.synchronize()
    assert state.step == 2 and all(np.isfinite(float(v)) for v in metrics.values())

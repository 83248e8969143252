"""Hyper-parameters read by the G+D training step.

Mirrors the fields of the reference's ``xmcgan/configs/coco_xmc.py:18-88`` that the hot path
reads ("Config fields read on the hot path").  ``ml_collections`` is not
available in this image, so ``ConfigDict`` is a small attribute-access dict with the same
``config.name`` / ``config["name"]`` surface.
"""
from __future__ import annotations


class ConfigDict(dict):
    """Attribute-access dict (the subset of ml_collections.ConfigDict the path uses)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def copy(self):
        return ConfigDict(self)


def get_config() -> ConfigDict:
    """Default 128 px configuration (reference coco_xmc.py:18-68)."""
    c = ConfigDict()
    c.seed = 42
    c.beta1 = 0.5
    c.beta2 = 0.999
    c.d_lr = 0.0004
    c.g_lr = 0.0001
    c.polyak_decay = 0.999
    c.batch_norm_group_size = -1
    c.dtype = "bfloat16"
    c.image_size = 128
    c.batch_size = 56
    c.eval_batch_size = 7
    c.df_dim = 96
    c.gf_dim = 96
    c.z_dim = 128
    c.d_step_per_g_step = 2
    c.g_spectral_norm = False
    c.d_spectral_norm = True
    c.architecture = "xmc_net"
    c.gamma_for_g = 15
    c.word_contrastive = True
    c.sentence_contrastive = True
    c.image_contrastive = True
    # coco_xmc.py:65: the frozen ResNet-50 image-contrastive term.  Its weights come from
    # ``pretrained_model_path`` (the reference's data/resnet_pretrained.npy; a network download -- absent here, the
    # create_additional_data raises FileNotFoundError like the reference; None = explicit random initialisation).
    c.pretrained_image_contrastive = True
    c.pretrained_model_path = "data/resnet_pretrained.npy"
    c.cond_size = 16
    c.show_num = 64                 # images per sampling grid (coco_xmc.py:42)
    # build-side switches (not in the reference)
    c.ema = True
    return c


def get_test_config() -> ConfigDict:
    """Tiny configuration C0 (reference coco_xmc.py:71-88: dims 16, z 8), per-device B=4."""
    c = get_config()
    c.dtype = "float32"
    c.pretrained_image_contrastive = False      # build-side: the G/D parity tests run without the ResNet-50 term;
    c.batch_size = 4                            # tests/test_resnet.py and tests/test_gpu_resnet.py switch it on
    c.eval_batch_size = 2
    c.show_num = 4                  # coco_xmc.py:85
    c.df_dim = 16
    c.gf_dim = 16
    c.z_dim = 8
    return c


def get_c1_config() -> ConfigDict:
    """BASELINE.json configs[1]: 128 px, gf=df=96, batch 56, bf16, EMA off."""
    c = get_config()
    c.ema = False
    return c


def get_c3_config() -> ConfigDict:
    """BASELINE.json configs[3]: 256 px paper config, per-GPU batch 32."""
    c = get_config()
    c.image_size = 256
    c.batch_size = 32
    c.ema = False
    return c


def get_c4_config() -> ConfigDict:
    """BASELINE.json configs[4]: the 256 px config with fp8 MFMA convolutions -- MX-fp8 operands (e4m3 + e8m0 per 32
    channels) for the 3x3 convolutions' forward and data-gradient passes, float32 accumulation; contrastive losses,
    attention, normalisation and weight gradients as in the bf16 mode (losses in float32)."""
    c = get_c3_config()
    c.conv_fp8 = True
    # scale of an MX block: "next_binade" (default; the OCP conversion, but one binade higher when the block maximum would saturate
    # e4m3) or "ocp_floor" (the OCP MX v1.0 conversion to the letter: clips the largest element of ~1/5 of the blocks by <= 12.5 %)
    c.fp8_scale_rule = "next_binade"
    return c

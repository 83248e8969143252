"""Step orchestration: ``TrainState``, ``create_train_state``, ``train_step``, ``eval_step``.

Mirrors the hot-path part of the reference's ``xmcgan/train_utils.py``: ``TrainState`` (:42-50),
``split_input_dict`` (:69-88), ``train_step`` (:91-130), ``create_train_state`` (:133-193).  The
reference has no ``eval_step``; the generator-only evaluation it performs in ``generate_batch``
(:245-309) / ``eval_metrics.py:90-124`` -- G(train=False) once with the parameters and once with
the EMA parameters -- is exposed here under that name.
"""
from __future__ import annotations

import dataclasses
import time
from typing import Any, Dict, Optional

import torch

from . import synthetic as syn
from . import xmc_gan
from .libml.layers import ParamArena
from .nets import xmc_net


class Optimizer:
    """flax.optim.Optimizer stand-in: ``.target`` is the Flax-layout parameter tree (views of the
    flat arena), ``.state`` exposes the Adam step and moment trees (train_utils.py:181-186)."""

    def __init__(self, arena: ParamArena, lr, beta1, beta2):
        self.arena, self.lr, self.beta1, self.beta2 = arena, lr, beta1, beta2
        self.target = arena.tree()

    @property
    def state(self):
        return {"step": self.arena.opt_step,
                "param_states": {"grad_ema": self.arena.tree(self.arena.m),
                                 "grad_sq_ema": self.arena.tree(self.arena.v)}}


@dataclasses.dataclass
class TrainState:
    """Data structure for checkpointing the model (reference train_utils.py:42-50)."""
    step: int
    g_optimizer: Optimizer
    d_optimizer: Optimizer
    generator_state: Optional[Any]
    discriminator_state: Optional[Any]
    ema_params: Any
    ema_buffer: Any = None          # flat arena behind ema_params (build-side)
    pending: Any = None             # deferred "wait for the D gradient exchange + Adam" of a train_d (multi-GPU only)
    prefetched_g: Any = None        # (batch identity, (image, new batch_stats, tape)) of the next train_g_d's generator forward (train_d)

    def replace(self, **kw):
        return dataclasses.replace(self, **kw)


class _NetFactory:
    """``functools.partial(generator_cls, config=..., dtype=...)`` of the reference
    (train_utils.py:159-161): called with ``train=`` it returns the (cached) network object."""

    def __init__(self, cls, config, dtype, ops):
        self.cls, self.config, self.dtype, self.ops = cls, config, dtype, ops
        self._cache = {}

    def __call__(self, train):
        if train not in self._cache:
            self._cache[train] = self.cls(self.config, train, dtype=self.dtype, ops=self.ops)
        return self._cache[train]


def split_input_dict(input_dict: Dict[str, torch.Tensor], splits: int, axis=0):
    """train_utils.py:69-88 -- zero-copy views."""
    out = [{} for _ in range(splits)]
    for k, v in input_dict.items():
        v = torch.as_tensor(v)
        if v.shape[axis] % splits != 0:              # jnp.split raises too; torch.chunk would silently give ragged parts
            raise ValueError(f"split_input_dict: '{k}' has {v.shape[axis]} rows, not divisible by {splits}")
        for i, part in enumerate(torch.chunk(v, splits, dim=axis)):
            out[i][k] = part
    return out


def create_train_state(config, rng, init_batch=None, ops=None):
    """-> (generator, discriminator, TrainState) (train_utils.py:133-193).  ``rng`` is an integer
    seed; G / D / (unused) z streams are derived from it like the 3-way split of the reference."""
    dtype = torch.bfloat16 if config.dtype == "bfloat16" else torch.float32
    xmc_net.check_config(config)
    ops = ops if ops is not None else xmc_net.make_ops(dtype)
    if config.get("conv_fp8", False):                # BASELINE config #5: MX-fp8 3x3 convolutions (ops.py::_conv_mx8)
        if dtype != torch.bfloat16:
            raise ValueError("config.conv_fp8 needs config.dtype = 'bfloat16' (activations between the fp8 convolutions are bf16)")
        ops.fp8 = True
        # scale rule of the MX quantisers: "next_binade" (default: no saturating block maximum) or "ocp_floor" (OCP MX v1.0 to the letter)
        ops.set_fp8_scale_rule(config.get("fp8_scale_rule", "next_binade"))
    generator = _NetFactory(xmc_net.Generator, config, dtype, ops)
    discriminator = _NetFactory(xmc_net.Discriminator, config, dtype, ops)
    seed = int(rng)
    g_vars = generator(train=False).init(seed, None)
    d_vars = discriminator(train=False).init(seed + 1, None)
    g_arena, d_arena = g_vars["params"].arena, d_vars["params"].arena
    ema_buffer = g_arena.params.clone()                                   # ema_params = generator_params (:170)
    state = TrainState(
        step=0,
        g_optimizer=Optimizer(g_arena, config.g_lr, config.beta1, config.beta2),
        d_optimizer=Optimizer(d_arena, config.d_lr, config.beta1, config.beta2),
        generator_state={"batch_stats": g_vars["batch_stats"]},
        discriminator_state={"spectral_norm_stats": d_vars["spectral_norm_stats"]},
        ema_params=g_arena.tree(ema_buffer), ema_buffer=ema_buffer)
    return generator, discriminator, state


def load_flax_params(state, g_params=None, g_batch_stats=None, d_params=None, d_sn_stats=None):
    """Install Flax-layout trees (numpy / torch leaves) into a TrainState (checkpoints, tests)."""
    ops = state.g_optimizer.arena.ops
    if g_params is not None:
        state.g_optimizer.arena.load_flax(g_params)
        state.ema_buffer.copy_(state.g_optimizer.arena.params)
    if d_params is not None:
        state.d_optimizer.arena.load_flax(d_params)
    new = {}
    if g_batch_stats is not None:
        new["generator_state"] = {"batch_stats": xmc_net._tree_to_dev(ops, g_batch_stats)}
    if d_sn_stats is not None:
        new["discriminator_state"] = {"spectral_norm_stats": xmc_net._tree_to_dev(ops, d_sn_stats)}
    return state.replace(**new)


def train_step(rng, state, batch, gan_model=xmc_gan, generator=None, discriminator=None, config=None,
               additional_data=None, grad_sync=None):
    """One G+D training step (train_utils.py:91-130): the per-device batch (leading dim
    B * d_step_per_g_step) is split; ``train_d`` runs on the first halves, ``train_g_d`` on the last."""
    n = config.d_step_per_g_step
    parts = split_input_dict(batch, n)
    rngs = [int(rng) * n + i for i in range(n)]      # one stream per half step (train_utils.py:121 splits the key)
    if (gan_model is xmc_gan and xmc_gan._RESNET_REAL_EARLY and grad_sync is None and n > 1 and additional_data
            and additional_data.get("image_model") is not None and config.get("pretrained_image_contrastive", False)):
        ops = generator(train=True).ops
        if hasattr(ops, "side") and ops.dtype == torch.bfloat16:
            xmc_gan.prefetch_pretrained_real(additional_data["image_model"], parts[-1], ops)
    for i in range(n - 1):
        # with replicas, train_d leaves its gradient exchange in flight: the D update is applied by the next half
        # step right before it first needs the D parameters (after train_g_d's generator forward)
        kw = {}
        if i == n - 2 and grad_sync is None and gan_model is xmc_gan:
            kw["next_g_batch"] = parts[-1]           # its generator forward runs beside this half step's backward
            kw["next_g_rng"] = rngs[-1]
            if additional_data and config.get("pretrained_image_contrastive", False):
                kw["next_image_model"] = additional_data.get("image_model")
        state = gan_model.train_d(rngs[i], state, parts[i], generator, discriminator, config, grad_sync=grad_sync,
                                  defer_update=grad_sync is not None, **kw)
    return gan_model.train_g_d(rngs[-1], state, parts[-1], generator, discriminator, config, additional_data or {},
                               grad_sync=grad_sync)


# hipStreamCaptureModeThreadLocal: only the capturing thread is held to the capture rules.  Under the default
# ("global") mode ANY thread's hipEventQuery aborts the capture -- and ProcessGroupNCCL's watchdog thread polls the
# end events of the eager collectives issued just before (every 100 ms, until it has seen them complete): a rare
# "operation not permitted when stream is capturing" + SIGABRT under torchrun (seen once in ~15 runs, round 3).
CAPTURE_ERROR_MODE = "thread_local"
WATCHDOG_DRAIN_S = 0.35


class GraphedTrainStep:
    """``train_step`` captured ONCE into a hipGraph and replayed.

    One eager ``train_step`` issues ~700 kernel launches through Python + ctypes (17-25 ms of host time); the
    captured graph is one ``hipGraphLaunch``.  What makes the step replayable:

    * every buffer a kernel touches is caller-owned (torch's caching allocator serves the capture from a private
      pool: the tape, workspaces and scratch of the step keep fixed addresses);
    * the batch lives in static tensors (``load_batch`` copies a new batch into them);
    * the Adam step counters live in device memory (``ParamArena.step_state``, xmc_adam_ema_dev);
    * parameters / moments / EMA are updated in place in their arenas, and the per-step state the reference
      returns as new pytrees -- G's BatchNorm running statistics, D's spectral-norm ``u0`` -- is copied back into
      the persistent flat buffers (``FlatTree``) at the end of the graph.

    Call the eager ``train_step`` at least once before constructing this (lazy per-device setup in the library,
    RCCL communicator creation), then ``state, metrics = graphed(state, batch)``.  ``state`` must be the object
    this instance returned last (or was built from): the graph owns the addresses of its tensors.
    """

    def __init__(self, state, batch, gan_model=xmc_gan, generator=None, discriminator=None, config=None,
                 additional_data=None, grad_sync=None):
        g, d = generator(train=True), discriminator(train=True)
        ops = g.ops
        dev = ops.device
        self.config = config
        if "z" not in batch:
            raise ValueError("GraphedTrainStep needs the noise in the batch (key 'z', coco_dataset.py:165-166): a "
                             "host-seeded draw inside train_d / train_g_d cannot be replayed by a captured graph")
        self.static_batch = {k: torch.as_tensor(v).to(dev).clone() for k, v in batch.items()}
        state = xmc_gan._flush(state)
        bs = g.flat_batch_stats(state.g_optimizer.target, state.generator_state["batch_stats"])
        sn = d.flat_sn_stats(state.d_optimizer.target, state.discriminator_state["spectral_norm_stats"])
        state = state.replace(generator_state={"batch_stats": bs}, discriminator_state={"spectral_norm_stats": sn})
        ga, da = state.g_optimizer.arena, state.d_optimizer.arena
        before = (ga.opt_step, da.opt_step, int(state.step))
        if additional_data and "image_model" in additional_data:
            additional_data["image_model"].bind(ops)     # the frozen ResNet-50's weights go to HBM before the capture
        torch.cuda.synchronize(dev)
        if grad_sync is not None:
            # ProcessGroupNCCL's watchdog keeps every eager collective in its list until it has SEEN it complete (it looks every
            # ~100 ms).  If the capture puts RCCL's stream into capture mode while such an entry is still there, the watchdog's
            # hipEventQuery on that (eagerly recorded) end event fails with "operation not permitted on an event last recorded
            # in a capturing stream" and the process aborts -- seen once in round 4 (profiles/r04: the D-exchange-in-one-piece
            # run).  All collectives are complete here (synchronize above): give the watchdog three of its periods to retire them.
            time.sleep(WATCHDOG_DRAIN_S)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, capture_error_mode=CAPTURE_ERROR_MODE):
            new_state, metrics = train_step(0, state, self.static_batch, gan_model, generator, discriminator, config,
                                            additional_data or {}, grad_sync=grad_sync)
            new_state = xmc_gan._flush(new_state)
            bs.flat.copy_(new_state.generator_state["batch_stats"].flat)
            sn.flat.copy_(new_state.discriminator_state["spectral_norm_stats"].flat)
        # capture executed nothing: undo the host-side mirrors the Python code advanced, keep what a replay must add
        self._d_steps, self._g_steps = da.opt_step - before[1], ga.opt_step - before[0]
        self._steps = int(new_state.step) - before[2]
        ga._opt_step, da._opt_step = before[0], before[1]
        ga.version += 1                                  # prepared-weight caches point into the graph's pool:
        da.version += 1                                  # any eager call after this must re-prepare
        self.state, self.metrics = state, metrics
        # ops.fuse_prep: the captured forward passes trust the prepared copies the previous step's optimiser kernel left in the
        # networks' persistent buffers; parameters changed behind the graph's back (load_flax_params, an eager step) are
        # detected by the arenas' version counters and re-prepared eagerly before the replay
        self._nets, self._versions = (g, d), (ga.version, da.version)

    def load_batch(self, batch):
        for k, dst in self.static_batch.items():
            src = torch.as_tensor(batch[k])
            if src.data_ptr() != dst.data_ptr():
                dst.copy_(src, non_blocking=True)

    def __call__(self, state=None, batch=None):
        """Replay one step.  ``batch`` (optional) is copied into the static input tensors first.
        -> (state, metrics): metrics are the graph's static output tensors (read them before the next replay)."""
        if state is not None and state is not self.state:
            raise ValueError("GraphedTrainStep replays on the state it returned last (the graph owns its buffers)")
        if batch is not None:
            self.load_batch(batch)
        st = self.state
        if (st.g_optimizer.arena.version, st.d_optimizer.arena.version) != self._versions:
            g, d = self._nets
            if hasattr(g, "refresh_prepared"):
                g.refresh_prepared(st.g_optimizer.target)
                d.refresh_prepared(st.d_optimizer.target, st.discriminator_state["spectral_norm_stats"])
        self.graph.replay()
        st.g_optimizer.arena.note_steps(self._g_steps)
        st.d_optimizer.arena.note_steps(self._d_steps)
        st.g_optimizer.arena.version += 1
        st.d_optimizer.arena.version += 1
        self._versions = (st.g_optimizer.arena.version, st.d_optimizer.arena.version)
        st.step += self._steps
        return st, self.metrics


def eval_step(rng, state, batch, generator, config):
    """Generator-only evaluation (train_utils.py:245-281, eval_metrics.py:90-124): images from the
    current and from the EMA parameters with running BatchNorm statistics; z ~ N(0, 1) from ``rng``
    unless the batch carries one."""
    if not config.get("ema", True) and int(state.step) > 0:
        import warnings
        warnings.warn("config.ema is False (the build-side switch of the C1 / C3 benchmark configs): the EMA parameters "
                      "were not updated by train_g_d and still hold their initial values", stacklevel=2)
    g = generator(train=False)
    cond = {k: batch[k] for k in ("sentence_embedding", "embedding", "max_len")}
    b = torch.as_tensor(batch["sentence_embedding"]).shape[0]
    if "z" in batch:
        z = batch["z"]
    else:
        gen = torch.Generator().manual_seed(int(rng))
        z = torch.randn((b, config.z_dim), generator=gen)
    variables = {"params": state.g_optimizer.target, **state.generator_state}
    image = g.apply(variables, (cond, z), mutable=False)
    ema_variables = {"params": state.ema_params, **state.generator_state}
    ema_image = g.apply(ema_variables, (cond, z), mutable=False)
    return image, ema_image


def generate_sample(rng, state, generator, config, cond=None, world_size=1):
    """Single-device sampling with a fresh ``z`` (reference train_utils.py:196-242) -> ``{"generated_image", "ema_generated_image"}``,
    each a ``make_grid`` of ``config.show_num`` images with the writer's leading [None] axis.  ``sample_size`` =
    ``config.batch_size // world_size`` (the reference divides by ``jax.device_count()``).

    The reference conditions on ``dict(sentence_embedding=one_hot(label, config.num_classes))`` -- a leftover of a class-conditional
    model: ``xmc_net.Generator`` reads ``embedding`` and ``max_len`` too (xmc_net.py:170-172), so the reference call raises
    ``KeyError('embedding')`` for every XMC config.  Called the same way (``cond=None``) this raises the same KeyError; with
    ``cond`` = a caption dict (``sentence_embedding``, ``embedding``, ``max_len``; leading dim >= sample_size) it samples."""
    from .utils import image_utils
    sample_size = config.batch_size // max(int(world_size), 1)
    gen = torch.Generator().manual_seed(int(rng))
    z = torch.randn((sample_size, config.z_dim), generator=gen)
    if cond is None:
        label = torch.randint(0, 1000, (sample_size,), generator=gen)
        cond = dict(sentence_embedding=torch.nn.functional.one_hot(label, config.get("num_classes", 1000)).float())
    missing = [k for k in ("embedding", "max_len") if k not in cond]
    if missing:
        raise KeyError(missing[0])
    cond = {k: torch.as_tensor(cond[k])[:sample_size] for k in ("sentence_embedding", "embedding", "max_len")}
    g = generator(train=False)
    image = g.apply({"params": state.g_optimizer.target, **state.generator_state}, (cond, z), mutable=False)
    ema_image = g.apply({"params": state.ema_params, **state.generator_state}, (cond, z), mutable=False)
    show = config.get("show_num", 64)
    return dict(generated_image=image_utils.make_grid(image.float(), show)[None],
                ema_generated_image=image_utils.make_grid(ema_image.float(), show)[None])


def generate_batch(rng, state, batch, generator, config, collect_all=False, group=None):
    """Sampling for visualisation / evaluation (reference train_utils.py:245-309): images from
    the current and from the EMA generator parameters (``train=False``: running BatchNorm statistics) and the
    originals, each as a ``make_grid`` of ``config.show_num`` float32 images.  ``collect_all`` concatenates the
    replicas' images first (``lax.all_gather`` of the reference -> ``torch.distributed.all_gather``)."""
    from .utils import image_utils
    image, ema_image = eval_step(rng, state, {k: v for k, v in batch.items() if k != "z"}, generator, config)
    ori = torch.as_tensor(batch["image"]).to(image.device)
    outs = [image.float(), ema_image.float(), ori.float()]
    if collect_all and torch.distributed.is_available() and torch.distributed.is_initialized():
        world = torch.distributed.get_world_size(group)
        gathered = []
        for t in outs:
            parts = [torch.empty_like(t) for _ in range(world)]
            torch.distributed.all_gather(parts, t.contiguous(), group=group)
            gathered.append(torch.cat(parts, dim=0))
        outs = gathered
    show = config.get("show_num", 64)
    # key names and the leading [None] axis of the reference's summary dict (train_utils.py:299-309)
    return {"generated_image_batch": image_utils.make_grid(outs[0], show)[None],
            "ema_generated_image_batch": image_utils.make_grid(outs[1], show)[None],
            "ori_image_batch": image_utils.make_grid(outs[2], show)[None]}

"""Data-parallel path on the HIP backend: two ranks share the one GPU of the test box over gloo (RCCL refuses two
ranks per device; the 8-GPU RCCL run is the driver's).  Exercises dp.GradSync on device arenas, the side-stream
exchange and the deferred discriminator update; ranks must end bit-identical on parameters."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_one_gpu_gloo():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29544", os.path.join(ROOT, "tools", "dp_smoke_one_gpu.py")]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "dp smoke OK" in out.stdout
    assert "g: finite=True identical_across_ranks=True" in out.stdout
    assert "d: finite=True identical_across_ranks=True" in out.stdout


def test_two_ranks_hip_backend_match_averaged_oracle(tmp_path):
    """SURVEY.md 8(a) a26 on the HIP backend: two float32 replicas (per-rank batches, gloo exchange of the DEVICE gradient
    arenas, the deferred discriminator update, the frozen ResNet-50 term on) against the single-process emulation that
    averages the two replicas' ORACLE gradients (lax.pmean, reference xmc_gan.py:170-171,251) and applies Adam:
    post-step parameters per leaf, and the replica-mean metrics (TrainMetrics, xmc_gan.py:185-190)."""
    import torch
    from oracle import torch_ref as R
    from tests.dp_reference import reference_two_replicas
    from tests.test_gpu_step import _noise_leaves
    from xmcgan_image_generation_amd import synthetic as syn
    from xmcgan_image_generation_amd.configs import coco_xmc
    from xmcgan_image_generation_amd.utils import resnet_v1
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", DP_DTYPE="float32", DP_DUMP=str(tmp_path))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29546", os.path.join(ROOT, "tools", "dp_smoke_one_gpu.py")]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    got = torch.load(os.path.join(tmp_path, "dp_rank0.pt"))
    cfg = coco_xmc.get_test_config()
    cfg.batch_size = 2
    cfg.pretrained_image_contrastive = True
    gp, gs = syn.init_generator(cfg, seed=42, bias_scale=0.05)
    dp_, ds = syn.init_discriminator(cfg, seed=43, bias_scale=0.05)
    resnet = resnet_v1.init_resnet50(7, head_scale=0.2)
    states, metrics, mean = reference_two_replicas(cfg, gp, gs, dp_, ds,
                                                   [syn.make_batch(cfg, per_device_batch=2, rank=r) for r in range(2)],
                                                   resnet=resnet)
    for tree_key, ref_key, gk in (("d_tree", "d_params", "d"), ("g_tree", "g_params", "g")):
        skip = _noise_leaves(R.leaves(mean[gk]))         # analytically-zero gradients: Adam turns round-off into +-lr steps
        worst, wp = 0.0, None
        for path, ref in R.leaves(states[0][ref_key]):
            if path in skip:
                continue
            a, b = got[tree_key][path].double(), ref.double()
            r = float((a - b).norm() / max(float(b.norm()), 1e-12)) if float(b.norm()) > 1e-6 else float((a - b).norm())
            if r > worst:
                worst, wp = r, path
        print(f"DP on HIP vs averaged oracle, {ref_key}: worst leaf norm-relative error {worst:.3e} at {wp}")
        assert worst < 1e-3, (ref_key, wp, worst)
    for k in ("d_loss", "g_loss", "c_loss_d", "c_loss_g", "c_loss_g_pretrained"):
        want = 0.5 * (float(metrics[0][k]) + float(metrics[1][k]))
        assert abs(got["metrics"][k] - want) <= 1e-3 * max(1.0, abs(want)), (k, got["metrics"][k], want)


def test_one_rank_rccl_backend():
    """The RCCL ("nccl") code path of dp.GradSync -- side-stream bucketed all-reduce of the device arenas, deferred
    discriminator update, replica-mean metrics -- with the one rank a 1-GPU box allows."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", DP_BACKEND="nccl")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", "29545", os.path.join(ROOT, "tools", "dp_smoke_one_gpu.py")]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "dp smoke OK" in out.stdout


def test_graph_capture_tolerates_the_rccl_watchdog_thread():
    """ProcessGroupNCCL's watchdog polls (hipEventQuery) the eager collectives issued just before a capture; under the
    default global capture mode such a query from ANY thread aborts the capture (seen once under torchrun in round 3).
    GraphedTrainStep captures with train_utils.CAPTURE_ERROR_MODE = thread_local: a 0.4 s capture right after an eager
    all-reduce, four times, with a captured all-reduce inside."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               MASTER_ADDR="127.0.0.1", MASTER_PORT="29571")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "capture_vs_watchdog.py")], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "capture vs watchdog OK" in out.stdout


@pytest.mark.parametrize("schedule", ["overlapped", "exclusive"])
def test_bench_under_torchrun_replays_the_graph_with_rccl_inside(schedule):
    """bench.py as the driver launches it for N > 1 (torch.distributed.run, RCCL backend; one rank here): the whole step
    incl. the RCCL all-reduces of both gradient arenas is captured into the hipGraph and replayed, finite losses -- under both
    exchange schedules (dp.GradSync.schedule)."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", "29547", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--config", "tiny",
           "--steps", "3", "--warmup", "2", "--graph", "on", "--no-cpu-baseline", "--no-instrument", "--grad-schedule", schedule]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["launch_mode"].startswith("hipGraph replay") and line["config"]["parallelism"] == "dp1"
    assert line["grad_schedule"]["chosen"] == schedule
    assert all(v == v and abs(v) < 1e6 for v in line["losses"].values()), line["losses"]


def test_bench_n2_control_flow_two_ranks_one_gpu():
    """bench.py --gpus 2 as the driver launches it, both ranks on the one GPU of the test box (gloo exchange: test hook
    XMC_BENCH_BACKEND): per-rank batches, barrier + max-over-ranks timing, whole-job images/sec (2 x batch per step),
    rank-0-only JSON line, finite replica-mean losses.  --grad-schedule auto (the default): both exchange schedules are timed
    during warm-up, every rank takes the same decision, the JSON line says which."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", XMC_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29548", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "tiny",
           "--steps", "3", "--warmup", "2", "--graph", "off", "--no-cpu-baseline", "--no-instrument"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines                                     # rank 0 only
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["parallelism"] == "dp2" and line["scaling"] == "weak"
    b = line["config"]["global_batch"]
    assert b == 2 * 4 and abs(line["value"] - b / (line["ms_per_step"] * 1e-3)) <= 1e-2 * line["value"]
    assert all(v == v and abs(v) < 1e6 for v in line["losses"].values()), line["losses"]
    gs = line["grad_schedule"]
    assert gs["chosen"] in ("overlapped", "exclusive") and set(gs["trial_ms_per_step"]) == {"overlapped", "exclusive"}, gs
    assert gs["chosen"] == min(gs["trial_ms_per_step"], key=gs["trial_ms_per_step"].get)


@pytest.mark.parametrize("pick", ["overlapped", "exclusive"])
def test_bench_schedule_trial_with_rccl_inside_the_graphs(pick):
    """The `--grad-schedule auto` trial of bench.py as it runs for N > 1 -- a hipGraph per exchange schedule (RCCL all-reduces inside),
    1 + 3 replays of each, then either the last graph is kept (pick = exclusive) or a fresh one is captured for the winner (pick =
    overlapped) -- forced at world size 1 on the RCCL backend (test hooks XMC_BENCH_FORCE_SCHEDULE_TRIAL / _PICK; _STRICT: a failure
    inside the trial fails the test instead of falling back)."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", XMC_BENCH_FORCE_SCHEDULE_TRIAL="1", XMC_BENCH_TRIAL_PICK=pick,
               XMC_BENCH_TRIAL_STRICT="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", "29549" if pick == "overlapped" else "29550", os.path.join(ROOT, "bench.py"), "--gpus", "1",
           "--config", "tiny", "--steps", "3", "--warmup", "2", "--graph", "on", "--no-cpu-baseline"] + \
        (["--no-instrument"] if pick == "overlapped" else [])        # exclusive: with the instrumented extra step, as the driver runs it
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert (line["roofline"] is None) == (pick == "overlapped")
    gs = line["grad_schedule"]
    assert gs["chosen"] == pick and set(gs["trial_ms_per_step"]) == {"overlapped", "exclusive"}, gs
    assert line["launch_mode"].startswith("hipGraph replay")
    assert all(v == v and abs(v) < 1e6 for v in line["losses"].values()), line["losses"]

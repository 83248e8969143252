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


def test_one_rank_rccl_backend():
    """The RCCL ("nccl") code path of dp.GradSync -- side-stream bucketed all-reduce of the device arenas, deferred
    discriminator update, replica-mean metrics -- with the one rank a 1-GPU box allows."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", DP_BACKEND="nccl")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", "29545", os.path.join(ROOT, "tools", "dp_smoke_one_gpu.py")]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "dp smoke OK" in out.stdout

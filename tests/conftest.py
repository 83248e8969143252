import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The whole session runs the PRODUCT optimiser mode (what bench.py times: the optimiser kernel consumes the gradient arena in
# place, ops.fuse_opt / ops.first_write).  Only the tests that READ a gradient arena after a step ask for the ``keep_grads``
# fixture: the kernel then writes the FINAL gradient (with the term through sigma) back and the next half step zero-fills
# (ops.keep_grads).  tests/test_gpu_fused_opt.py holds the two modes to bit-identical parameters.
os.environ.pop("XMC_KEEP_GRADS", None)


@pytest.fixture
def keep_grads(monkeypatch):
    """HipOps reads XMC_KEEP_GRADS when it is constructed: operator tables built inside the test keep the final gradient"""
    monkeypatch.setenv("XMC_KEEP_GRADS", "1")
    yield


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run via gpurun)")


@pytest.fixture(scope="session")
def tiny_cfg():
    from xmcgan_image_generation_amd.configs import coco_xmc
    return coco_xmc.get_test_config()

"""The C ABI from plain C: tests/c/abi_client.c is compiled with gcc against include/xmcgan_hip.h and linked with
libxmcgan_hip.so -- version handshake on CPU, one float32 convolution through the ABI on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "xmcgan_image_generation_amd")


def _build(tmp_path):
    exe = str(tmp_path / "abi_client")
    cmd = ["gcc", "-O1", "-std=c11", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "abi_client.c"),
           "-o", exe, "-L", PKG, "-lxmcgan_hip", "-L/opt/rocm/lib", "-lamdhip64", "-lm",
           f"-Wl,-rpath,{PKG}", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


def test_c_client_links_and_handshakes(tmp_path):
    exe = _build(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.startswith("abi ")


@pytest.mark.gpu
def test_c_client_runs_a_convolution(tmp_path):
    exe = _build(tmp_path)
    out = subprocess.run([exe, "gpu"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "conv max abs err" in out.stdout

"""The C-ABI shared library loads and exports every symbol include/xmcgan_hip.h declares; the
ctypes signature table covers exactly that set.  No GPU needed (no compute calls)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "xmcgan_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\bint(?:64_t)?\s+(xmc_\w+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from xmcgan_image_generation_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build it first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/xmcgan_hip.h but not exported"
    assert lib.xmc_abi_version() == _lib.ABI_VERSION


def test_ctypes_table_matches_header():
    from xmcgan_image_generation_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared()
    _lib.load()


def test_product_path_fails_loudly_without_gpu():
    """No CPU fallback: constructing the HIP operator table without a GPU raises."""
    import pytest
    import torch
    from xmcgan_image_generation_amd import _lib
    from xmcgan_image_generation_amd.ops import HipOps
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.XmcError):
        HipOps()


def test_product_package_never_imports_oracle():
    pkg = os.path.join(ROOT, "xmcgan_image_generation_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                assert "cpu_ops" not in txt, f

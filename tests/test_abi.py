"""CPU: the C-ABI library loads and exports exactly the symbols include/zkast.h declares; no compute calls."""
import ctypes
import os
import re

import pytest

from zkast import lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "zkast.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(zk_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def handle():
    if not os.path.exists(lib.LIB_PATH):
        lib.build()
    return lib.load_library()


def test_header_and_binding_agree():
    assert _declared() == sorted(lib.SYMBOLS)


def test_every_declared_symbol_is_exported(handle):
    for name in _declared():
        assert hasattr(handle, name), name
    assert handle.zk_version().decode().startswith("zkast")


def test_no_cpu_fallback(handle):
    """without a GPU zk_create must fail loudly; with one this test is skipped (the gpu suite covers it)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = ctypes.c_void_p()
    rc = handle.zk_create(0, ctypes.byref(h))
    assert rc != 0 and not h.value
    assert b"no HIP device" in handle.zk_last_error(None) or b"HIP" in handle.zk_last_error(None)
    with pytest.raises(lib.ZkError):
        lib.Context(0)


def test_product_path_never_imports_oracle():
    """oracle/ is the checker: nothing under the package (the product path) may import, load or execute it."""
    pkg = os.path.join(ROOT, "zenker-audio-detection_amd")
    pat = re.compile(r"^\s*(from|import)\s+oracle\b|oracle[/.]ast_oracle|import_module\(.oracle", re.M)
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp", ".sh")):
                text = open(os.path.join(dirpath, fn)).read()
                assert not pat.search(text), (dirpath, fn)

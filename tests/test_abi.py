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


def test_library_has_no_hard_wired_hip_runtime():
    """libzkast.so must not carry DT_NEEDED libamdhip64 / a RUNPATH into one ROCm tree: it binds to the process's runtime
    (zkast/lib.py::_ensure_hip_runtime), otherwise a torch wheel's own runtime makes two (round 3: "No HIP GPUs")."""
    import subprocess
    if not os.path.exists(lib.LIB_PATH):
        lib.build()
    dyn = subprocess.run(["readelf", "-d", lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "libamdhip64" not in dyn and "RUNPATH" not in dyn and "RPATH" not in dyn, dyn


def test_one_hip_runtime_image_whatever_the_import_order():
    """fresh processes, both orders: after zkast AND torch are loaded exactly one libamdhip64 and one libhsa-runtime64
    image is mapped (no GPU needed: loading is enough)."""
    import subprocess
    import sys
    body = (
        "import os, sys\n"
        f"sys.path.insert(0, {os.path.join(ROOT, 'zenker-audio-detection_amd')!r})\n"
        "{first}\n{second}\n"
        "from zkast import lib\n"
        "hip, hsa = lib._mapped_libraries('libamdhip64'), lib._mapped_libraries('libhsa-runtime64')\n"
        "assert len(hip) == 1 and len(hsa) <= 1, (hip, hsa)\n"
        "assert os.path.samefile(hip[0], lib.HIP_RUNTIME_PATH) or not lib.HIP_RUNTIME_PATH.startswith('/'), (hip, lib.HIP_RUNTIME_PATH)\n"
        "print('ok', hip[0])\n")
    zk = "from zkast import lib; lib.load_library()"
    th = "import torch"
    for first, second in ((zk, th), (th, zk)):
        r = subprocess.run([sys.executable, "-c", body.format(first=first, second=second)], capture_output=True, text=True,
                           timeout=300)
        assert r.returncode == 0 and r.stdout.startswith("ok"), (first, r.stdout, r.stderr[-2000:])


def _run_py(code, env_extra):
    import subprocess
    import sys
    env = dict(os.environ, **env_extra)
    env["PYTHONPATH"] = os.path.join(ROOT, "zenker-audio-detection_amd") + os.pathsep + env.get("PYTHONPATH", "")
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)


def test_rccl_override_that_does_not_load_is_an_error_with_a_reason(handle):
    """$ZKAST_RCCL_LIB is an explicit override: a file that cannot be loaded must come back as an error code and a message
    (round 4: the path crashed on a second dlerror() call) — and must not fall through to some other RCCL."""
    r = _run_py("from zkast import lib\n"
                "try:\n    lib.comm_unique_id()\n    print('LOADED')\n"
                "except lib.ZkError as e:\n    print('ZKERROR', e)\n", {"ZKAST_RCCL_LIB": "/nonexistent/librccl-missing.so"})
    assert r.returncode == 0, r.stderr[-2000:]
    assert "ZKERROR" in r.stdout and "/nonexistent/librccl-missing.so" in r.stdout and "LOADED" not in r.stdout, r.stdout


def test_hip_runtime_override_wins_or_fails_loudly(handle):
    """$ZKAST_HIP_LIB ranks first (round 4 ranked it below a torch wheel's runtime); a path that does not load raises."""
    r = _run_py("from zkast import lib\n"
                "try:\n    lib.load_library()\n    print('LOADED', lib.HIP_RUNTIME_PATH)\n"
                "except lib.ZkError as e:\n    print('ZKERROR', e)\n", {"ZKAST_HIP_LIB": "/nonexistent/libamdhip64.so"})
    assert r.returncode == 0, r.stderr[-2000:]
    assert "ZKERROR" in r.stdout and "ZKAST_HIP_LIB=/nonexistent/libamdhip64.so" in r.stdout, r.stdout
    sysrt = "/opt/rocm/lib/libamdhip64.so.7"
    if os.path.exists(sysrt):
        r = _run_py("from zkast import lib\nlib.load_library()\nprint('LOADED', lib.HIP_RUNTIME_PATH)\n", {"ZKAST_HIP_LIB": sysrt})
        assert r.returncode == 0 and f"LOADED {sysrt}" in r.stdout, (r.stdout, r.stderr[-2000:])
        # another runtime already mapped: refuse instead of loading a second one
        import importlib.util
        spec = importlib.util.find_spec("torch")
        wheel_rt = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so") if spec and spec.origin else ""
        if os.path.exists(wheel_rt) and os.path.realpath(wheel_rt) != os.path.realpath(sysrt):
            r = _run_py(f"import ctypes\nctypes.CDLL({wheel_rt!r}, mode=ctypes.RTLD_GLOBAL)\nfrom zkast import lib\n"
                        "try:\n    lib.load_library()\n    print('LOADED')\n"
                        "except lib.ZkError as e:\n    print('ZKERROR', e)\n", {"ZKAST_HIP_LIB": sysrt})
            assert "ZKERROR" in r.stdout and "already mapped" in r.stdout, (r.stdout, r.stderr[-2000:])


def test_default_mix_assignment_is_stated_once():
    """ZK_F16MIX's default assignment lives in csrc/zkast.hip (ZK_MIX_X3_MASK: one nibble per layer — QKV GEMM, QK^T, O projection,
    MLP) and is mirrored by zkast.lib.MIX_X3_GROUPS (what bench.py and the docs print): the two must say the same."""
    src = open(os.path.join(ROOT, "zenker-audio-detection_amd", "csrc", "zkast.hip")).read()
    mask = int(re.search(r"#define ZK_MIX_X3_MASK (0x[0-9a-fA-F]+)ull", src).group(1), 16)
    want = {}
    for layer in range(12):
        g = (mask >> (4 * layer)) & 15
        if g:
            want[layer] = tuple(name for bit, name in enumerate(lib.LAYER_GROUPS) if g >> bit & 1)
    assert want == {k: tuple(v) for k, v in lib.MIX_X3_GROUPS.items()}
    modes = lib.mix_layer_modes()
    assert len(modes) == 12 and all(len(m) == 4 for m in modes)
    assert modes[0] == ("f16x3", "f16x3", "f16c8", "f16c8") and all(m == ("f16c8",) * 4 for m in modes[1:])
    assert lib.mix_layer_modes({3: ("mlp",)})[3] == ("f16c8", "f16c8", "f16c8", "f16x3")
    assert lib.DEFAULT_COMPUTE_MODE in lib.COMPUTE_MODES and lib.COMPUTE_MODES["f16mix"] == lib.ZK_F16MIX == 4
    hdr = open(os.path.join(ROOT, "include", "zkast.h")).read()
    assert re.search(r"ZK_F16MIX\s*=\s*4", hdr)

"""ctypes binding of libzkast.so (the C ABI in include/zkast.h).

This is the whole boundary between the Python host code and the HIP kernels: plain pointers and sizes, no torch
types.  There is NO CPU fallback: if the shared library is missing or no gfx950 GPU is visible every entry point
raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# ZKAST_LIB: alternative build of the same library (kernel experiments under tools/); default is the in-tree build.
# The library is linked -no-hip-rt (no DT_NEEDED on libamdhip64): open it through load_library(), or after a HIP runtime is
# in the global scope (_ensure_hip_runtime) — a bare dlopen in a process without one fails on unresolved hip* symbols.
LIB_PATH = os.environ.get("ZKAST_LIB") or os.path.join(_HERE, "libzkast.so")
CSRC = os.path.join(os.path.dirname(_HERE), "csrc")

ZK_F16, ZK_F16C8, ZK_F16X3, ZK_F16MIX = 1, 2, 3, 4
ZK_DT_F32, ZK_DT_F16, ZK_DT_BF16 = 0, 1, 2
EPI_STORE, EPI_GELU, EPI_RESID, EPI_PATCH = 0, 1, 2, 3
TEST_TILED_IN, TEST_TILED_OUT, TEST_POISON_PAD, TEST_SHORT_X = 0x100, 0x200, 0x400, 0x800      # include/zkast.h: ZK_TEST_*
# the kernel groups ZK_F16MIX runs as ZK_F16X3 (everything else ZK_F16C8), {layer: groups}: ZK_MIX_X3_MASK of csrc/zkast.hip
LAYER_GROUPS = ("qkv", "att", "o", "mlp")      # fused QKV GEMM, QK^T of attention, O projection, MLP (FC1 + FC2)
MIX_X3_GROUPS = {0: ("qkv", "att")}


def mix_layer_modes(x3_groups=None, n_layers: int = 12) -> list:
    """per-layer (qkv, att, o, mlp) mode tuples of a ZK_F16MIX assignment: f16x3 for the groups named in `x3_groups`
    ({layer: groups}, default MIX_X3_GROUPS), f16c8 elsewhere"""
    g = MIX_X3_GROUPS if x3_groups is None else x3_groups
    return [tuple("f16x3" if k in g.get(l, ()) else "f16c8" for k in LAYER_GROUPS) for l in range(n_layers)]
# what the drop-in classes, the CLIs and bench.py use unless told otherwise: the cheapest mode that keeps >= 20 % of the 1e-3
# logit tolerance on a configs[3]-sized recording of the input-sensitive weight set (tests/test_sens_tail_gpu.py)
DEFAULT_COMPUTE_MODE = "f16mix"
COMPUTE_MODES = {"f16": ZK_F16, "f16c8": ZK_F16C8, "f16x3": ZK_F16X3, "f16mix": ZK_F16MIX, 1: ZK_F16, 2: ZK_F16C8, 3: ZK_F16X3, 4: ZK_F16MIX}

# every symbol include/zkast.h declares (tests/test_abi.py checks the .so exports exactly these)
SYMBOLS = [
    "zk_create", "zk_destroy", "zk_last_error", "zk_set_stream", "zk_set_async", "zk_synchronize",
    "zk_set_micro_batch", "zk_set_prune_last_layer", "zk_set_layer0_reuse", "zk_set_layer0_attention", "zk_version", "zk_model_load", "zk_model_set_compute_mode", "zk_model_set_layer_modes", "zk_model_set_fx",
    "zk_logmel", "zk_features_expand", "zk_features_get", "zk_features_set", "zk_ast_forward", "zk_softmax", "zk_two_stage", "zk_gate",
    "zk_comm_unique_id", "zk_comm_init", "zk_comm_destroy", "zk_comm_info", "zk_allgather_logits", "zk_comm_allgather_bytes",
    "zk_resample", "zk_wav_decode", "zk_audio_load", "zk_audio_get", "zk_prof_begin", "zk_prof_end", "zk_prof_get", "zk_prof_get_flops", "zk_debug_set_tap", "zk_debug_get_tap",
    "zk_test_layernorm", "zk_test_gemm", "zk_test_attention", "zk_test_split_c8",
]


class ZkError(RuntimeError):
    pass


class TensorDesc(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("ndim", C.c_int32), ("shape", C.c_int64 * 4),
                ("dtype", C.c_int32)]


class ASTConfigC(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("hidden_size", "num_hidden_layers", "num_attention_heads",
                                         "intermediate_size", "patch_size", "frequency_stride", "time_stride",
                                         "max_length", "num_mel_bins", "num_labels")] + [("layer_norm_eps", C.c_float)]


def build(verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into libzkast.so (hipcc cross-compiles without a GPU)."""
    r = subprocess.run(["bash", os.path.join(CSRC, "build.sh")], capture_output=True, text=True)
    if verbose or r.returncode:
        print(r.stdout, r.stderr)
    if r.returncode:
        raise ZkError("building libzkast.so failed:\n" + r.stderr[-4000:])
    return LIB_PATH


_lib = None
_lib_lock = threading.Lock()
HIP_RUNTIME_PATH = None      # the libamdhip64 this process's libzkast.so is bound to (set by load_library)


def _mapped_libraries(prefix: str) -> list:
    """paths of the shared objects mapped into this process whose file name starts with `prefix`"""
    seen = []
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                path = line.rstrip("\n").split(None, 5)[-1] if line.count("/") else ""
                if path.startswith("/") and os.path.basename(path).startswith(prefix) and path not in seen:
                    seen.append(path)
    except OSError:
        pass
    return seen


def _ensure_hip_runtime() -> str:
    """Exactly ONE HIP runtime per process.  libzkast.so is linked without a DT_NEEDED on libamdhip64 (csrc/build.sh): its
    hip* symbols resolve against whichever runtime is in the global scope when it is loaded, and this function puts one
    there, in this order:

      0. ``$ZKAST_HIP_LIB`` when set: the explicit override wins.  If ANOTHER libamdhip64 is already mapped the call raises
         instead of loading a second runtime beside it;
      1. a libamdhip64 that is already mapped — the host imported torch first, or is a HIP application itself;
      2. the one a PyTorch wheel ships (``torch/lib/libamdhip64.so``, located WITHOUT importing torch): a later
         ``import torch`` then finds its runtime already loaded and shares it;
      3. the system ROCm (``libamdhip64.so.7`` on the loader path, ``/opt/rocm/lib``).

    A loader that bypasses this function (a C host, another binding) must put a HIP runtime into the global scope itself
    before it opens libzkast.so: the library is linked ``-no-hip-rt`` and has no DT_NEEDED that would do it.

    Why: a PyTorch-ROCm wheel carries its own SONAME-less ``libamdhip64.so`` + ``libhsa-runtime64.so``; with libzkast.so
    hard-wired to /opt/rocm's the process held two HIP and two HSA runtimes, and the one that came up second could not
    acquire the GPU (``RuntimeError: No HIP GPUs are available`` from torch after libzkast had run — round 3,
    tests/test_model_gpu.py on its own).  The order of imports no longer matters."""
    mode = getattr(os, "RTLD_GLOBAL", 0x100) | getattr(os, "RTLD_NOW", 0x2)
    tried = []
    cands = _mapped_libraries("libamdhip64.so")
    override = os.environ.get("ZKAST_HIP_LIB")
    if override:
        other = [p for p in cands if os.path.realpath(p) != os.path.realpath(override)]
        if other:
            raise ZkError(f"ZKAST_HIP_LIB={override} but another HIP runtime is already mapped into this process ({other[0]}): "
                          "two HIP runtimes cannot share a GPU; unset ZKAST_HIP_LIB or load libzkast before that runtime")
        try:
            C.CDLL(override, mode=mode)
            return override
        except OSError as e:
            raise ZkError(f"ZKAST_HIP_LIB={override} could not be loaded: {e}") from e
    if not cands:
        try:
            import importlib.util
            spec = importlib.util.find_spec("torch")
            if spec is not None and spec.origin:
                p = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
                if os.path.exists(p):
                    cands.append(p)
        except (ImportError, ValueError):
            pass
    cands += ["libamdhip64.so.7", "/opt/rocm/lib/libamdhip64.so.7", "libamdhip64.so"]
    for path in cands:
        try:
            C.CDLL(path, mode=mode)      # an already-mapped file is not loaded again, only promoted to the global scope
            return path
        except OSError as e:
            tried.append(f"{path}: {e}")
    raise ZkError("no HIP runtime (libamdhip64) could be loaded; zkast has no CPU fallback.  Tried: " + "; ".join(tried))


def load_library() -> C.CDLL:
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise ZkError(f"{LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(zkast has no CPU fallback; the HIP library is the product)")
        global HIP_RUNTIME_PATH
        HIP_RUNTIME_PATH = _ensure_hip_runtime()
        lib = C.CDLL(LIB_PATH)
        vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
        sig = {
            "zk_create": (C.c_int, [C.c_int, C.POINTER(vp)]),
            "zk_destroy": (None, [vp]),
            "zk_last_error": (C.c_char_p, [vp]),
            "zk_set_stream": (C.c_int, [vp, vp]),
            "zk_set_async": (C.c_int, [vp, C.c_int]),
            "zk_synchronize": (C.c_int, [vp]),
            "zk_set_micro_batch": (C.c_int, [vp, i32]),
            "zk_set_prune_last_layer": (C.c_int, [vp, C.c_int]),
            "zk_set_layer0_reuse": (C.c_int, [vp, C.c_int]),
            "zk_set_layer0_attention": (C.c_int, [vp, C.c_int]),
            "zk_version": (C.c_char_p, []),
            "zk_model_load": (C.c_int, [vp, C.c_int, C.POINTER(TensorDesc), i32, C.POINTER(ASTConfigC), f32, f32, i32]),
            "zk_model_set_compute_mode": (C.c_int, [vp, C.c_int, i32]),
            "zk_model_set_layer_modes": (C.c_int, [vp, C.c_int, C.POINTER(i32), i32]),
            "zk_model_set_fx": (C.c_int, [vp, C.c_int, f32, f32]),
            "zk_logmel": (C.c_int, [vp, vp, i64, i64, i64, i32, i32]),
            "zk_features_expand": (C.c_int, [vp, f32, f32, i32, vp]),
            "zk_features_get": (C.c_int, [vp, vp, C.POINTER(i32), C.POINTER(i32)]),
            "zk_features_set": (C.c_int, [vp, vp, i32, i32]),
            "zk_ast_forward": (C.c_int, [vp, C.c_int, vp, vp, i32, vp]),
            "zk_softmax": (C.c_int, [vp, vp, i32, i32, vp]),
            "zk_two_stage": (C.c_int, [vp, vp, i64, i64, i64, i32, i32, f32, f32, vp, vp, vp, vp]),
            "zk_gate": (C.c_int, [vp, vp, i32, f32, f32, vp, vp, vp]),
            "zk_comm_unique_id": (C.c_int, [vp]),
            "zk_comm_init": (C.c_int, [vp, i32, i32, vp]),
            "zk_comm_destroy": (C.c_int, [vp]),
            "zk_comm_info": (C.c_int, [vp, C.POINTER(i32), C.POINTER(i32)]),
            "zk_allgather_logits": (C.c_int, [vp, vp, i32, i32, vp]),
            "zk_comm_allgather_bytes": (C.c_int, [vp, vp, i64, vp]),
            "zk_resample": (C.c_int, [vp, vp, i64, i32, i32, vp, i64]),
            "zk_audio_load": (C.c_int, [vp, vp, i64, i32, i32, i32, i32, i32, C.POINTER(i64)]),
            "zk_audio_get": (C.c_int, [vp, vp, C.POINTER(i64)]),
            "zk_prof_begin": (C.c_int, [vp]),
            "zk_prof_end": (C.c_int, [vp]),
            "zk_prof_get": (C.c_int, [vp, C.c_char_p, C.POINTER(C.c_double), C.POINTER(i64)]),
            "zk_prof_get_flops": (C.c_int, [vp, C.c_char_p, C.POINTER(C.c_double)]),
            "zk_debug_set_tap": (C.c_int, [vp, i32]),
            "zk_debug_get_tap": (C.c_int, [vp, vp, i32]),
            "zk_wav_decode": (C.c_int, [vp, vp, i64, i32, i32, i32, vp]),
            "zk_test_layernorm": (C.c_int, [vp, vp, vp, vp, i32, f32, i32, vp]),
            "zk_test_split_c8": (C.c_int, [vp, vp, C.c_int64, i32, i32, vp]),
            "zk_test_gemm": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp]),
            "zk_test_attention": (C.c_int, [vp, vp, i32, i32, vp]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


def comm_unique_id() -> bytes:
    """128-byte RCCL unique id (rank 0 creates it and ships it to the other ranks over any host channel)."""
    buf = C.create_string_buffer(128)
    rc = load_library().zk_comm_unique_id(buf)
    if rc:
        why = load_library().zk_last_error(None)
        raise ZkError(f"zk_comm_unique_id failed ({rc}): {why.decode() if why else 'RCCL (librccl.so.1) could not be loaded'}")
    return buf.raw


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


def _ptr(x):
    """(address, keepalive) of a C-contiguous numpy array or torch tensor (host or device)."""
    if x is None:
        return None, None
    if _is_torch(x):
        if not x.is_contiguous():
            x = x.contiguous()
        return x.data_ptr(), x
    a = np.ascontiguousarray(x)
    return a.ctypes.data, a


PROF_CLASSES = ["gemm_qkv", "gemm_o", "gemm_fc1", "gemm_fc2", "gemm_patch", "attention", "layernorm", "logmel",
                "embed", "head", "wav_decode", "resample", "allgather",
                "gemm_qkv_x3", "gemm_o_x3", "gemm_fc1_x3", "gemm_fc2_x3", "attention_x3"]


class Context:
    """One (process, GPU) context: weights of the two stages, the feature slot, workspace and a HIP stream."""

    def __init__(self, device: int = 0):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.zk_create(int(device), C.byref(h))
        if rc:
            raise ZkError(f"zk_create({device}) failed ({rc}): {self.lib.zk_last_error(None).decode()}")
        self.h = h
        self.device = int(device)
        self.stage_labels = {}

    def close(self):
        if getattr(self, "h", None):
            self.lib.zk_destroy(self.h)
            self.h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc:
            raise ZkError(f"{what} failed ({rc}): {self.lib.zk_last_error(self.h).decode()}")

    # ---- configuration ----
    def set_micro_batch(self, windows: int):
        self._chk(self.lib.zk_set_micro_batch(self.h, int(windows)), "zk_set_micro_batch")

    def set_prune_last_layer(self, enable: bool):
        self._chk(self.lib.zk_set_prune_last_layer(self.h, int(bool(enable))), "zk_set_prune_last_layer")

    def set_layer0_reuse(self, enable: bool):
        self._chk(self.lib.zk_set_layer0_reuse(self.h, int(bool(enable))), "zk_set_layer0_reuse")

    def set_layer0_attention(self, enable: bool):
        self._chk(self.lib.zk_set_layer0_attention(self.h, int(bool(enable))), "zk_set_layer0_attention")

    def set_stream(self, hip_stream: int | None):
        self._chk(self.lib.zk_set_stream(self.h, C.c_void_p(hip_stream or 0)), "zk_set_stream")

    def set_async(self, enable: bool):
        self._chk(self.lib.zk_set_async(self.h, int(bool(enable))), "zk_set_async")

    def synchronize(self):
        self._chk(self.lib.zk_synchronize(self.h), "zk_synchronize")

    # ---- model ----
    def load_model(self, stage: int, state_dict: dict, config: dict, fx_mean: float, fx_std: float, compute_mode=ZK_F16C8):
        descs = (TensorDesc * len(state_dict))()
        keep = []
        for i, (name, arr) in enumerate(state_dict.items()):
            if _is_torch(arr):
                import torch
                t = arr.detach().cpu().contiguous()
                if t.dtype == torch.bfloat16:
                    a, dt = t.view(torch.int16).numpy(), ZK_DT_BF16
                elif t.dtype == torch.float16:
                    a, dt = t.numpy(), ZK_DT_F16
                else:
                    a, dt = t.float().numpy(), ZK_DT_F32
            else:
                a = np.ascontiguousarray(arr)
                if a.dtype == np.float16:
                    dt = ZK_DT_F16
                else:
                    a, dt = np.ascontiguousarray(a, dtype=np.float32), ZK_DT_F32
            nb = name.encode()
            keep.append((a, nb))
            d = descs[i]
            d.name = nb
            d.data = a.ctypes.data
            d.ndim = min(a.ndim, 4) if a.ndim <= 4 else 4
            shp = list(a.shape) if a.ndim <= 4 else [int(np.prod(a.shape[:-3]))] + list(a.shape[-3:])
            for j in range(4):
                d.shape[j] = shp[j] if j < len(shp) else 1
            d.dtype = dt
        cfg = ASTConfigC()
        defaults = dict(hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                        patch_size=16, frequency_stride=10, time_stride=10, max_length=1024, num_mel_bins=128,
                        num_labels=2)
        for k, v in defaults.items():
            setattr(cfg, k, int(config.get(k, v)))
        cfg.layer_norm_eps = float(config.get("layer_norm_eps", 1e-12))
        mode = COMPUTE_MODES[compute_mode]
        self._chk(self.lib.zk_model_load(self.h, int(stage), descs, len(state_dict), C.byref(cfg), float(fx_mean),
                                         float(fx_std), mode), f"zk_model_load(stage={stage})")
        self.stage_labels[int(stage)] = int(cfg.num_labels)

    def set_compute_mode(self, stage: int, mode):
        self._chk(self.lib.zk_model_set_compute_mode(self.h, int(stage), COMPUTE_MODES[mode]), "zk_model_set_compute_mode")

    def set_layer_modes(self, stage: int, modes):
        """one compute mode per encoder layer ("f16c8" / "f16x3" / "f16"), or one (qkv, att, o, mlp) tuple of modes per
        layer; the model's mode becomes f16mix"""
        per_group = any(isinstance(m, (tuple, list)) for m in modes)
        flat = []
        for m in modes:
            if isinstance(m, (tuple, list)):
                if len(m) != 4:
                    raise ValueError("a per-layer entry is one mode or a (qkv, att, o, mlp) tuple of modes")
                flat += [COMPUTE_MODES[v] for v in m]
            else:
                flat += [COMPUTE_MODES[m]] * (4 if per_group else 1)
        arr = (C.c_int32 * len(flat))(*flat)
        self._chk(self.lib.zk_model_set_layer_modes(self.h, int(stage), arr, len(flat)), "zk_model_set_layer_modes")

    def set_fx(self, stage: int, mean: float, std: float):
        self._chk(self.lib.zk_model_set_fx(self.h, int(stage), float(mean), float(std)), "zk_model_set_fx")

    # ---- features ----
    def logmel(self, audio, n_samples: int, first_start: int, hop: int, win: int, n_windows: int):
        p, _k = _ptr(audio)
        self._chk(self.lib.zk_logmel(self.h, p, int(n_samples), int(first_start), int(hop), int(win), int(n_windows)),
                  "zk_logmel")

    def features_expand(self, mean: float, std: float, do_normalize: bool, out):
        p, _k = _ptr(out)
        self._chk(self.lib.zk_features_expand(self.h, float(mean), float(std), int(bool(do_normalize)), p),
                  "zk_features_expand")

    def features_shape(self):
        nw, nf = C.c_int32(), C.c_int32()
        self._chk(self.lib.zk_features_get(self.h, None, C.byref(nw), C.byref(nf)), "zk_features_get")
        return nw.value, nf.value

    def features_get(self) -> np.ndarray:
        nw, nf = self.features_shape()
        out = np.empty((nw, nf, 128), dtype=np.float32)
        if nw:
            self._chk(self.lib.zk_features_get(self.h, out.ctypes.data, None, None), "zk_features_get")
        return out

    def features_set(self, feats):
        """compact un-normalised log-mel (N, n_frames, 128) float32, numpy or device tensor -> the feature slot"""
        n, nf, nm = (int(v) for v in feats.shape)
        if nm != 128:
            raise ValueError(f"features must have 128 mel bins, got {nm}")
        if not _is_torch(feats):
            feats = np.ascontiguousarray(feats, dtype=np.float32)
        p, _k = _ptr(feats)
        self._chk(self.lib.zk_features_set(self.h, p, n, nf), "zk_features_set")

    # ---- transformer ----
    def ast_forward(self, stage: int, input_values, win_idx, B: int, logits_out):
        pi, _k1 = _ptr(input_values)
        px, _k2 = _ptr(win_idx)
        po, _k3 = _ptr(logits_out)
        self._chk(self.lib.zk_ast_forward(self.h, int(stage), pi, px, int(B), po), "zk_ast_forward")

    def softmax(self, logits: np.ndarray) -> np.ndarray:
        logits = np.ascontiguousarray(logits, dtype=np.float32)
        out = np.empty_like(logits)
        if logits.size:
            self._chk(self.lib.zk_softmax(self.h, logits.ctypes.data, logits.shape[0], logits.shape[1], out.ctypes.data),
                      "zk_softmax")
        return out

    def gate(self, logits: np.ndarray, thr1: float, fwd_min_prob=None):
        logits = np.ascontiguousarray(logits, dtype=np.float32)
        n = logits.shape[0]
        probs = np.empty((n, 2), np.float32)
        idx = np.empty((max(n, 1),), np.int32)
        cnt = C.c_int32(0)
        self._chk(self.lib.zk_gate(self.h, logits.ctypes.data, n, float(thr1),
                                   -1.0 if fwd_min_prob is None else float(fwd_min_prob), probs.ctypes.data,
                                   idx.ctypes.data, C.addressof(cnt)), "zk_gate")
        return probs, idx[: cnt.value].copy()

    def two_stage(self, audio, n_samples, first_start, hop, win, n_windows, thr1, fwd_min_prob=None):
        pa, _k = _ptr(audio)
        n = int(n_windows)
        s1 = np.empty((n, 2), np.float32)
        s2 = np.empty((max(n, 1), 2), np.float32)
        idx = np.empty((max(n, 1),), np.int32)
        cnt = C.c_int32(0)
        self._chk(self.lib.zk_two_stage(self.h, pa, int(n_samples), int(first_start), int(hop), int(win), n,
                                        float(thr1), -1.0 if fwd_min_prob is None else float(fwd_min_prob),
                                        s1.ctypes.data, idx.ctypes.data, C.addressof(cnt), s2.ctypes.data),
                  "zk_two_stage")
        k = cnt.value
        return s1, idx[:k].copy(), s2[:k].copy()

    def two_stage_into(self, audio, n_samples, first_start, hop, win, n_windows, thr1, fwd_min_prob, s1_logits,
                       swallow_idx, n_swallow, s2_logits):
        """zk_two_stage with caller-provided outputs (numpy or torch, host or device): s1_logits (N,2), swallow_idx (N)
        int32, n_swallow (>=1) int32, s2_logits (N,2).  Nothing but the gate count crosses PCIe when all are device."""
        pa, _k0 = _ptr(audio)
        p1, _k1 = _ptr(s1_logits)
        pi, _k2 = _ptr(swallow_idx)
        pc, _k3 = _ptr(n_swallow)
        p2, _k4 = _ptr(s2_logits)
        self._chk(self.lib.zk_two_stage(self.h, pa, int(n_samples), int(first_start), int(hop), int(win), int(n_windows),
                                        float(thr1), -1.0 if fwd_min_prob is None else float(fwd_min_prob),
                                        p1, pi, pc, p2), "zk_two_stage")

    def resample(self, audio: np.ndarray, orig_sr: int, new_sr: int) -> np.ndarray:
        audio = np.ascontiguousarray(audio, dtype=np.float32)
        g = int(np.gcd(orig_sr, new_sr))
        o, nw = orig_sr // g, new_sr // g
        n_out = (nw * audio.shape[0] + o - 1) // o
        out = np.empty((n_out,), np.float32)
        self._chk(self.lib.zk_resample(self.h, audio.ctypes.data, audio.shape[0], int(orig_sr), int(new_sr),
                                       out.ctypes.data, n_out), "zk_resample")
        return out

    def resample_into(self, audio, n_in: int, orig_sr: int, new_sr: int, out):
        """zk_resample on caller-provided buffers (numpy or device tensors); out holds ceil(new*n_in/orig) samples."""
        pi, _k1 = _ptr(audio)
        po, _k2 = _ptr(out)
        self._chk(self.lib.zk_resample(self.h, pi, int(n_in), int(orig_sr), int(new_sr), po, int(out.shape[0])),
                  "zk_resample")

    def wav_decode(self, raw: bytes, format_tag: int, bits: int, channels: int) -> np.ndarray:
        """sample bytes of a WAVE data chunk -> mono float32 (channel mean), decoded on the GPU"""
        buf = np.frombuffer(raw, dtype=np.uint8)
        n_frames = buf.size // (channels * (bits // 8))
        out = np.empty((n_frames,), np.float32)
        if n_frames:
            self._chk(self.lib.zk_wav_decode(self.h, buf.ctypes.data, buf.size, int(format_tag), int(bits), int(channels),
                                             out.ctypes.data), "zk_wav_decode")
        return out

    def audio_load(self, raw: bytes, format_tag: int, bits: int, channels: int, sr: int, target_sr: int) -> int:
        """load_audio on the device: one upload of the data chunk, decode + channel mean + resample; the recording
        stays in the context's audio slot (two_stage / logmel with audio=None).  Returns its length in samples."""
        buf = np.frombuffer(raw, dtype=np.uint8)
        n = C.c_int64(0)
        self._chk(self.lib.zk_audio_load(self.h, buf.ctypes.data, buf.size, int(format_tag), int(bits), int(channels),
                                         int(sr), int(target_sr), C.byref(n)), "zk_audio_load")
        return n.value

    def audio_len(self) -> int:
        n = C.c_int64(0)
        self._chk(self.lib.zk_audio_get(self.h, None, C.byref(n)), "zk_audio_get")
        return n.value

    def audio_get(self) -> np.ndarray:
        out = np.empty((self.audio_len(),), np.float32)
        if out.size:
            self._chk(self.lib.zk_audio_get(self.h, out.ctypes.data, None), "zk_audio_get")
        return out

    # ---- multi-GPU (RCCL behind the C ABI) ----
    def comm_init(self, rank: int, world: int, unique_id: bytes | None):
        """Collective: bind an RCCL communicator to this context (world 1 needs no id and no RCCL)."""
        buf = C.create_string_buffer(unique_id, 128) if unique_id is not None else None
        self._chk(self.lib.zk_comm_init(self.h, int(rank), int(world), buf), "zk_comm_init")

    def comm_destroy(self):
        self._chk(self.lib.zk_comm_destroy(self.h), "zk_comm_destroy")

    def comm_info(self):
        r, w = C.c_int32(), C.c_int32()
        self._chk(self.lib.zk_comm_info(self.h, C.byref(r), C.byref(w)), "zk_comm_info")
        return r.value, w.value

    def allgather_logits(self, local, rows_per_rank: int, cols: int, out):
        """local (rows_per_rank, cols) fp32 -> out (world, rows_per_rank, cols); numpy or torch, host or device."""
        pl_, _k1 = _ptr(local)
        po, _k2 = _ptr(out)
        self._chk(self.lib.zk_allgather_logits(self.h, pl_, int(rows_per_rank), int(cols), po), "zk_allgather_logits")

    def allgather_bytes(self, payload: bytes) -> list:
        """Every rank contributes `payload` (lengths may differ); returns the list of all ranks' payloads."""
        _r, world = self.comm_info()
        n = np.array([len(payload)], np.int64)
        lens = np.zeros((world,), np.int64)
        self._chk(self.lib.zk_comm_allgather_bytes(self.h, n.ctypes.data, 8, lens.ctypes.data), "zk_comm_allgather_bytes")
        width = int(lens.max())
        if width == 0:
            return [b""] * world
        send = np.zeros((width,), np.uint8)
        send[: len(payload)] = np.frombuffer(payload, np.uint8)
        recv = np.zeros((world, width), np.uint8)
        self._chk(self.lib.zk_comm_allgather_bytes(self.h, send.ctypes.data, width, recv.ctypes.data), "zk_comm_allgather_bytes")
        return [recv[r, : int(lens[r])].tobytes() for r in range(world)]

    # ---- measurement ----
    def prof_begin(self):
        self._chk(self.lib.zk_prof_begin(self.h), "zk_prof_begin")

    def prof_end(self) -> dict:
        self._chk(self.lib.zk_prof_end(self.h), "zk_prof_end")
        out = {}
        for name in PROF_CLASSES:
            ms, n = C.c_double(), C.c_int64()
            self._chk(self.lib.zk_prof_get(self.h, name.encode(), C.byref(ms), C.byref(n)), "zk_prof_get")
            fl = C.c_double()
            self._chk(self.lib.zk_prof_get_flops(self.h, name.encode(), C.byref(fl)), "zk_prof_get_flops")
            out[name] = (ms.value, n.value, fl.value)
        return out

    def debug_tap(self, layer: int):
        self._chk(self.lib.zk_debug_set_tap(self.h, int(layer)), "zk_debug_set_tap")

    def debug_get_tap(self, n_windows: int) -> np.ndarray:
        out = np.empty((n_windows, 1214, 768), np.float32)
        self._chk(self.lib.zk_debug_get_tap(self.h, out.ctypes.data, int(n_windows)), "zk_debug_get_tap")
        return out

    # ---- single-kernel test hooks ----
    def test_layernorm(self, x, gamma, beta, eps, nsplit, tiled=False):
        x = np.ascontiguousarray(x, np.float32)
        out = np.empty_like(x)
        self._chk(self.lib.zk_test_layernorm(self.h, x.ctypes.data, np.ascontiguousarray(gamma, np.float32).ctypes.data,
                                             np.ascontiguousarray(beta, np.float32).ctypes.data, x.shape[0], float(eps),
                                             int(nsplit) | (TEST_TILED_OUT if tiled else 0), out.ctypes.data),
                  "zk_test_layernorm")
        return out

    def test_gemm(self, x, w, bias, epi, nsplit, resid=None, pos=None, tiled_in=False, tiled_out=False, poison_pad=False,
                  short_x=False):
        x = np.ascontiguousarray(x, np.float32)
        w = np.ascontiguousarray(w, np.float32)
        bias = np.ascontiguousarray(bias, np.float32)
        M, K = x.shape
        N = w.shape[0]
        if epi == EPI_PATCH:
            out = np.ascontiguousarray(resid, np.float32).copy()
            pos = np.ascontiguousarray(pos, np.float32)
        elif epi == EPI_RESID:
            out = np.ascontiguousarray(resid, np.float32).copy()
        else:
            out = np.empty((M, N), np.float32)
        flags = (int(epi) | (TEST_TILED_IN if tiled_in else 0) | (TEST_TILED_OUT if tiled_out else 0)
                 | (TEST_POISON_PAD if poison_pad else 0) | (TEST_SHORT_X if short_x else 0))
        self._chk(self.lib.zk_test_gemm(self.h, x.ctypes.data, w.ctypes.data, bias.ctypes.data, M, N, K, flags,
                                        int(nsplit), None if pos is None else pos.ctypes.data, out.ctypes.data),
                  "zk_test_gemm")
        return out

    def test_split_c8(self, x, w_exp=0, is_weight=False):
        x = np.ascontiguousarray(x, np.float32).reshape(-1)
        out = np.empty(x.size, np.uint16)
        self._chk(self.lib.zk_test_split_c8(self.h, x.ctypes.data, x.size, int(w_exp), int(bool(is_weight)),
                                            out.ctypes.data), "zk_test_split_c8")
        return out

    def test_attention(self, qkv, n_windows, nsplit, tiled=False):
        qkv = np.ascontiguousarray(qkv, np.float32)
        out = np.empty((qkv.shape[0], 768), np.float32)
        self._chk(self.lib.zk_test_attention(self.h, qkv.ctypes.data, int(n_windows),
                                             int(nsplit) | (TEST_TILED_OUT if tiled else 0), out.ctypes.data),
                  "zk_test_attention")
        return out


_contexts: dict = {}


def get_context(device: int = 0) -> Context:
    """Process-wide context per GPU (the reference's module-level DEVICE, src/test_long_audio_windows_2stage.py:48)."""
    ctx = _contexts.get(device)
    if ctx is None or ctx.h is None:
        ctx = Context(device)
        _contexts[device] = ctx
    return ctx

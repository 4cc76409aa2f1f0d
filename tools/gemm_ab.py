"""A/B timing of the ZK_F16C8 GEMM variants on the production shapes, device-resident random operands, interleaved
rounds in ONE process (libzkast_probes.so: `ZK_PROBES=1 csrc/build.sh`), plus a bit-exact comparison of the outputs.
usage: python tools/gemm_ab.py [windows=512] [variants=3] [iters=4] [rounds=5]"""
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__))); from _hip import cdll as _hip_cdll
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = _hip_cdll(os.environ.get("ZKAST_PROBES", os.path.join(ROOT, "zenker-audio-detection_amd", "zkast", "libzkast_probes.so")))
lib.zkp_bench_gemm_c8.restype = C.c_int
lib.zkp_bench_gemm_c8.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_float), C.POINTER(C.c_ulonglong)]

windows = int(sys.argv[1]) if len(sys.argv) > 1 else 512
variants = int(sys.argv[2]) if len(sys.argv) > 2 else 3
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 4
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 5
M = windows * 1214
shapes = [("qkv", 2304, 768, 0), ("fc1", 3072, 768, 1), ("o", 768, 768, 2), ("fc2", 768, 3072, 2)]
only = os.environ.get("AB_ONLY")
for name, N, K, epi in shapes:
    if only and name not in only.split(","):
        continue
    ms = (C.c_float * 2)()
    mm = C.c_ulonglong(0)
    rc = lib.zkp_bench_gemm_c8(M, N, K, epi, variants, iters, rounds, ms, C.byref(mm))
    if rc:
        raise SystemExit(f"{name}: probe failed rc={rc}")
    fl = 2.0 * M * N * K
    parts = [f"v{v + 1} {ms[v]:8.3f} ms {fl / (ms[v] * 1e-3) / 1e12:7.1f} TFLOP/s" for v in range(2) if variants & (1 << v)]
    extra = f"  speedup {ms[0] / ms[1]:.3f}x  mismatching dwords {mm.value}" if variants == 3 else ""
    print(f"{name:4s} M={M} N={N} K={K}: " + " | ".join(parts) + extra, flush=True)

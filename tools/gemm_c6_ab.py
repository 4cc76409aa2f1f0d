"""A/B of the production ZK_F16C8 GEMM against the MX-fp6 correction-plane variant (csrc/gemm_c6.hip, probes library only)
on the production shapes: ms per launch, algorithmic TFLOP/s, and how far the two outputs are apart.
usage: python tools/gemm_c6_ab.py [windows=512] [iters=4] [rounds=3]"""
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__))); from _hip import cdll as _hip_cdll
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = _hip_cdll(os.environ.get("ZKAST_PROBES", os.path.join(ROOT, "zenker-audio-detection_amd", "zkast", "libzkast_probes.so")))
lib.zkp_bench_gemm_c6.restype = C.c_int
lib.zkp_bench_gemm_c6.argtypes = [C.c_int] * 6 + [C.POINTER(C.c_float), C.POINTER(C.c_float)]
windows = int(sys.argv[1]) if len(sys.argv) > 1 else 512
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
M = windows * 1214
shapes = [("qkv", 2304, 768, 0), ("fc1", 3072, 768, 1), ("o", 768, 768, 2), ("fc2", 768, 3072, 2)]
only = os.environ.get("AB_ONLY")
for name, N, K, epi in shapes:
    if only and name not in only.split(","):
        continue
    ms = (C.c_float * 2)()
    err = (C.c_float * 2)()
    rc = lib.zkp_bench_gemm_c6(M, N, K, epi, iters, rounds, ms, err)
    if rc:
        raise SystemExit(f"{name}: probe failed rc={rc}")
    fl = 2.0 * M * N * K
    print(f"{name:4s} M={M} N={N} K={K}: c8 {ms[0]:7.3f} ms {fl / ms[0] / 1e9:6.1f} TF | c6 {ms[1]:7.3f} ms {fl / ms[1] / 1e9:6.1f} TF"
          f"  speedup {ms[0] / ms[1]:.3f}x   max|c6-c8| {err[0]:.3e}  (max|out| {err[1]:.3e})", flush=True)

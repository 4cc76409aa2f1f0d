"""Fill-path probe: the ZK_F16C8 GEMM's LDS-DMA stream alone (libzkast_probes.so), TB/s and GB/s per CU.
usage: python tools/fill_probe.py [windows=512]"""
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__))); from _hip import cdll as _hip_cdll
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = _hip_cdll(os.environ.get("ZKAST_PROBES", os.path.join(ROOT, "zenker-audio-detection_amd", "zkast", "libzkast_probes.so")))
lib.zkp_fill_probe.restype = C.c_int
lib.zkp_fill_probe.argtypes = [C.c_int] * 6 + [C.POINTER(C.c_float)]
M = (int(sys.argv[1]) if len(sys.argv) > 1 else 512) * 1214
for name, N, K in [("qkv", 2304, 768), ("fc1", 3072, 768), ("o", 768, 768), ("fc2", 768, 3072)]:
    for depth in (1, 2):
        for fix in (0, 1, 2, 3, 4):
            ms = C.c_float()
            rc = lib.zkp_fill_probe(M, N, K, depth, fix, 3, C.byref(ms))
            if rc:
                raise SystemExit(rc)
            tiles = ((M + 255) // 256) * (N // 256)
            byts = tiles * (K // 64) * 2 * 65536
            print(f"{name:4s} depth {depth} fix {fix}: {ms.value:7.3f} ms  {byts / ms.value / 1e9:6.2f} TB/s  "
                  f"{byts / ms.value / 1e6 / 256:6.1f} GB/s per CU", flush=True)

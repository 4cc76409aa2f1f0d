"""attention alone (libzkast_probes*.so given by ZKAST_PROBES): ms per launch and algorithmic TFLOP/s."""
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__))); from _hip import cdll as _hip_cdll
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = _hip_cdll(os.environ.get("ZKAST_PROBES", os.path.join(ROOT, "zenker-audio-detection_amd", "zkast", "libzkast_probes.so")))
lib.zkp_bench_attention.restype = C.c_int
lib.zkp_bench_attention.argtypes = [C.c_int] * 4 + [C.POINTER(C.c_float)]
W = int(sys.argv[1]) if len(sys.argv) > 1 else 512
for ns in (3, 2, 1):
    ms = C.c_float()
    rc = lib.zkp_bench_attention(W, ns, 3, 5, C.byref(ms))
    fl = W * 12 * 4.0 * 1214 * 1214 * 64
    print(f"attention nsplit {ns}: {ms.value:7.3f} ms  {fl / ms.value / 1e9:7.1f} TFLOP/s", flush=True)

"""Interleaved timing of several builds of the attention kernel (probe libraries from tools/build_variant.sh ... attention):
every round visits every build once, median over rounds.
usage: python tools/attn_ab_multi.py <windows> <rounds> <name>[,<name>...]   (name "base" = libzkast_probes.so)"""
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__))); from _hip import cdll as _hip_cdll
import ctypes as C
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ZK = os.path.join(ROOT, "zenker-audio-detection_amd", "zkast")


def load(name):
    lib = _hip_cdll(os.path.join(ZK, "libzkast_probes.so" if name in ("", "base") else f"libzkast_probes_{name}.so"))
    lib.zkp_bench_attention.restype = C.c_int
    lib.zkp_bench_attention.argtypes = [C.c_int] * 4 + [C.POINTER(C.c_float)]
    return lib


W = int(sys.argv[1]) if len(sys.argv) > 1 else 512
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
names = (sys.argv[3] if len(sys.argv) > 3 else "base").split(",")
libs = [load(n) for n in names]
fl = W * 12 * 4.0 * 1214 * 1214 * 64
for ns in (2, 3):
    t = [[] for _ in libs]
    for _ in range(rounds):
        for i, lib in enumerate(libs):
            ms = C.c_float()
            if lib.zkp_bench_attention(W, ns, 3, 1, C.byref(ms)):
                raise SystemExit(f"{names[i]}: probe failed")
            t[i].append(ms.value)
    med = [statistics.median(x) for x in t]
    for i, n in enumerate(names):
        print(f"nsplit {ns} {n:10s} median {med[i]:7.3f} ms  min {min(t[i]):7.3f}  {fl / med[i] / 1e9:6.1f} TFLOP/s  "
              f"x{med[0] / med[i]:.3f} vs {names[0]}", flush=True)

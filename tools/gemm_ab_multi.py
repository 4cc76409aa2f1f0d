"""Interleaved timing of SEVERAL builds of the ZK_F16C8 GEMM (probe libraries made by tools/build_variant.sh) on the
production shapes, in ONE process: every round visits every build once (cdna_hip_programming.md §5.4 rule 24), random
device-resident operands, median over rounds.  The first build is the reference of the speed-up column; each build is
also compared dword for dword against the frozen round-1 kernel inside its own probe library.
usage: python tools/gemm_ab_multi.py <windows> <rounds> <name>[,<name>...]     (name "" or "base" = libzkast_probes.so)
env AB_ONLY=qkv,fc1,o,fc2   AB_ITERS=4   AB_CHECK=0 (skip the bit comparison)"""
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__))); from _hip import cdll as _hip_cdll
import ctypes as C
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ZK = os.path.join(ROOT, "zenker-audio-detection_amd", "zkast")


def load(name):
    path = os.path.join(ZK, "libzkast_probes.so" if name in ("", "base") else f"libzkast_probes_{name}.so")
    lib = _hip_cdll(path)
    lib.zkp_bench_gemm_c8.restype = C.c_int
    lib.zkp_bench_gemm_c8.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_float), C.POINTER(C.c_ulonglong)]
    return lib


def main():
    windows = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    names = (sys.argv[3] if len(sys.argv) > 3 else "base").split(",")
    iters = int(os.environ.get("AB_ITERS", "4"))
    libs = [load(n) for n in names]
    M = windows * 1214
    shapes = [("qkv", 2304, 768, 0), ("fc1", 3072, 768, 1), ("o", 768, 768, 2), ("fc2", 768, 3072, 2)]
    only = os.environ.get("AB_ONLY")
    for sname, N, K, epi in shapes:
        if only and sname not in only.split(","):
            continue
        t = [[] for _ in libs]
        for _ in range(rounds):
            for i, lib in enumerate(libs):
                ms = (C.c_float * 2)()
                mm = C.c_ulonglong(0)
                rc = lib.zkp_bench_gemm_c8(M, N, K, epi, 2, iters, 1, ms, C.byref(mm))
                if rc:
                    raise SystemExit(f"{sname}/{names[i]}: probe failed rc={rc}")
                t[i].append(ms[1])
        fl = 2.0 * M * N * K
        med = [statistics.median(x) for x in t]
        for i, n in enumerate(names):
            line = (f"{sname:4s} {n or 'base':12s} median {med[i]:7.3f} ms  min {min(t[i]):7.3f}  "
                    f"{fl / (med[i] * 1e-3) / 1e12:6.1f} TFLOP/s  x{med[0] / med[i]:.3f} vs {names[0] or 'base'}")
            if os.environ.get("AB_CHECK", "1") != "0":
                ms = (C.c_float * 2)()
                mm = C.c_ulonglong(0)
                libs[i].zkp_bench_gemm_c8(M, N, K, epi, 3, 1, 1, ms, C.byref(mm))
                line += f"  dwords differing from the round-1 kernel: {mm.value}"
            print(line, flush=True)


if __name__ == "__main__":
    main()

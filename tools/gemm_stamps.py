"""Per-phase stamps of the ZK_F16C8 GEMM's ring steps (probe builds with -DZK_C8_STAMPS=1, tools/build_variant.sh):
three s_memtime stamps per step and wave, A = all MFMAs / LDS-DMA pieces of the step issued, B = the next step's data
landed (vmcnt(0)), C = step barrier released.  Prints, per build / shape / step kind / wave half, the medians of
   issue = A - C(previous step)    wait = B - A    barrier = C - B    step = C - C(previous step)      [shader cycles]
over the last 64 steps of every (workgroup, wave); steps that carry a tile epilogue (> 1.6 x the median step) are left out.
Also prints the clock the chip held during the launch (s_memtime against the 100 MHz s_memrealtime, per workgroup).
usage: python tools/gemm_stamps.py <windows> <name>[,<name>...]      (libzkast_probes_<name>.so)"""
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__))); from _hip import cdll as _hip_cdll
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ZK = os.path.join(ROOT, "zenker-audio-detection_amd", "zkast")


def main():
    windows = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    names = sys.argv[2].split(",")
    M = windows * 1214
    shapes = [("qkv", 2304, 768, 0), ("fc1", 3072, 768, 1), ("o", 768, 768, 2), ("fc2", 768, 3072, 2)]
    if os.environ.get("ZKP_SHAPES"):      # e.g. ZKP_SHAPES=fc1,qkv
        shapes = [sh for sh in shapes if sh[0] in os.environ["ZKP_SHAPES"].split(",")]
    clocks = {}
    for n in names:
        lib = _hip_cdll(os.path.join(ZK, f"libzkast_probes_{n}.so"))
        lib.zkp_bench_gemm_c8.restype = C.c_int
        lib.zkp_bench_gemm_c8.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_float), C.POINTER(C.c_ulonglong)]
        lib.zkp_c8_stamps_read.restype = C.c_int
        buf = np.zeros(256 * 8 * 64 * 4, np.uint32)
        for sname, N, K, epi in shapes:
            ms = (C.c_float * 2)()
            mm = C.c_ulonglong(0)
            if lib.zkp_bench_gemm_c8(M, N, K, epi, 2, 40, 1, ms, C.byref(mm)):      # 80 back-to-back launches: the clock has settled
                raise SystemExit("probe failed")
            if lib.zkp_c8_stamps_read(buf.ctypes.data_as(C.POINTER(C.c_uint))):
                raise SystemExit("stamp read failed")
            clk = np.zeros(256 * 4, np.uint64)
            lib.zkp_c8_clock_read(clk.ctypes.data_as(C.POINTER(C.c_ulonglong)))
            ck = clk.reshape(256, 4).astype(np.float64)
            ghz = np.median((ck[:, 2] - ck[:, 0]) / np.maximum(ck[:, 3] - ck[:, 1], 1.0)) * 0.1
            e = buf.reshape(256, 8, 64, 4).astype(np.int64)
            rows = {}      # (kind, half) -> list of (issue, wait, bar, step)
            for b in range(256):
                for w in range(8):
                    x = e[b, w]
                    x = x[np.argsort(x[:, 3])]
                    ids = x[:, 3]
                    ok = (ids[1:] - ids[:-1] > 0) & (ids[1:] // 2 - ids[:-1] // 2 == 1)      # consecutive steps
                    a, bb, c = x[1:, 0], x[1:, 1], x[1:, 2]
                    cp = x[:-1, 2]
                    d = lambda u, v: (u - v) & 0xFFFFFFFF
                    issue, wait, bar, step = d(a, cp), d(bb, a), d(c, bb), d(c, cp)
                    kind = ids[1:] & 1
                    for k in (0, 1):
                        sel = ok & (kind == k)
                        if sel.any():
                            rows.setdefault((k, w // 4), []).append(np.stack([issue[sel], wait[sel], bar[sel], step[sel]], 1))
            clocks.setdefault(n, {})["gemm_" + sname] = {"ghz": round(float(ghz), 3), "ms_per_launch": round(float(ms[1]), 3)}
            if not buf.any():      # -DZK_C8_STAMPS=2: clock stamps only
                print(f"{n} {sname}: kernel {ms[1]:.3f} ms, in-kernel clock {ghz:.2f} GHz (no per-step stamps in this build)", flush=True)
                continue
            print(f"{n} {sname} (kernel {ms[1]:.3f} ms with the stamps; in-kernel clock {ghz:.2f} GHz = d s_memtime / d s_memrealtime x 100 MHz, median over the workgroups)")
            for (k, h), v in sorted(rows.items()):
                v = np.concatenate(v)
                med = np.median(v[:, 3])
                v = v[v[:, 3] < 1.6 * med]
                q = np.median(v, 0)
                print(f"   {'fp16' if k == 0 else 'c8  '} step, waves {'0-3' if h == 0 else '4-7'}: issue {q[0]:6.0f}  wait {q[1]:5.0f}  "
                      f"barrier {q[2]:5.0f}  step {q[3]:6.0f}   ({len(v)} steps)", flush=True)
    out = os.environ.get("CLOCK_JSON")
    if out:      # profiles/r04_gemm_clock.json: what bench.py quotes as roofline.in_kernel_clock_ghz
        import hashlib
        import json
        first = names[0]
        so = os.path.join(ZK, "libzkast.so")
        json.dump({"note": f"tools/gemm_stamps.py {windows} {first}: probe build -DZK_C8_STAMPS=2 (s_memtime / s_memrealtime pair at kernel "
                           "entry and exit, the k-loop runs as shipped), 80 back-to-back launches per shape on random operands, "
                           "median over the 256 workgroups; clock = d s_memtime / d s_memrealtime x 100 MHz",
                   "windows": windows, "libzkast_sha256_at_measurement": hashlib.sha256(open(so, "rb").read()).hexdigest(),
                   "kernels": clocks[first]}, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()

"""GEMM probe (GPU box): production shapes of one 107-window micro-batch in ZK_F16X3 (3) vs ZK_F16C8 (2).
usage: rocprofv3 --kernel-trace --stats -d gpurun_out/probe -- python3 tools/gemm_probe_c8.py   (ZK_GEMM_STAMPS=1 for clocks)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "zenker-audio-detection_amd"))
import numpy as np
from zkast import lib
ctx = lib.get_context(0)
rng = np.random.default_rng(0)
M = 107 * 1214
for (N, K, epi) in [(2304, 768, lib.EPI_STORE), (768, 768, lib.EPI_RESID), (3072, 768, lib.EPI_GELU), (768, 3072, lib.EPI_RESID)]:
    x = rng.normal(0, 1, (M, K)).astype(np.float32)
    w = rng.normal(0, 0.05, (N, K)).astype(np.float32)
    b = np.zeros(N, np.float32)
    r0 = np.zeros((M, N), np.float32) if epi == lib.EPI_RESID else None
    outs = {}
    for ns in ((2,) if os.environ.get('ONLY_C8') else (3, 2)):
        for _ in range(2):
            outs[ns] = ctx.test_gemm(x, w, b, epi, ns, resid=None if r0 is None else r0.copy())
    d = np.abs(outs[2] - outs[3]).max() / np.abs(outs[3]).max() if 3 in outs else -1.0
    print(f"N={N} K={K} epi={epi}: c8 vs x3 max rel-to-scale diff {d:.2e}", flush=True)
print("done")

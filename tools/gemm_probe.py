"""GEMM micro-probe (GPU box): time zk_test_gemm shapes under rocprofv3 --kernel-trace to separate per-k-step
cost from per-tile cost.  usage: rocprofv3 --kernel-trace --stats ... -- python3 tools/gemm_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "zenker-audio-detection_amd"))
import numpy as np
from zkast import lib
ctx = lib.get_context(0)
rng = np.random.default_rng(0)
for (M, N, K) in [(8192, 768, 768), (8192, 768, 3072), (65536, 768, 768), (65536, 768, 3072), (65536, 3072, 768)]:
    x = rng.normal(0, 1, (M, K)).astype(np.float32)
    w = rng.normal(0, 0.05, (N, K)).astype(np.float32)
    b = np.zeros(N, np.float32)
    for ns in (1, 3):
        for _ in range(3):
            ctx.test_gemm(x, w, b, lib.EPI_STORE, ns)
print("done")

"""Time the c8 GEMM (ticks per ring step via ZK_GEMM_STAMPS) for the library named by ZKAST_LIB; results are not checked
(ablation builds compute garbage on purpose)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "zenker-audio-detection_amd"))
import numpy as np
from zkast import lib
ctx = lib.get_context(0)
rng = np.random.default_rng(0)
M = 107 * 1214
shapes = [(768, 3072, lib.EPI_RESID), (2304, 768, lib.EPI_STORE)]
if os.environ.get("AB_ALL"):
    shapes += [(768, 768, lib.EPI_RESID), (3072, 768, lib.EPI_GELU)]
for (N, K, epi) in shapes:
    x = rng.normal(0, 1, (M, K)).astype(np.float32)
    w = rng.normal(0, 0.05, (N, K)).astype(np.float32)
    b = np.zeros(N, np.float32)
    r0 = np.zeros((M, N), np.float32) if epi == lib.EPI_RESID else None
    for _ in range(2):
        ctx.test_gemm(x, w, b, epi, 2, resid=r0)

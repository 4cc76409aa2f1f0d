import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "zenker-audio-detection_amd"))
import numpy as np
from zkast import lib
ctx = lib.get_context(0)
rng = np.random.default_rng(0)
shapes = [(8192, 768, 3072)] if len(sys.argv) < 2 else [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for (M, N, K) in shapes:
    x = rng.normal(0, 1, (M, K)).astype(np.float32)
    w = rng.normal(0, 0.05, (N, K)).astype(np.float32)
    b = np.zeros(N, np.float32)
    for ns in (1, 3):
        for _ in range(2):
            ctx.test_gemm(x, w, b, lib.EPI_STORE, ns)

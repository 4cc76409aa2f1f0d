import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "zenker-audio-detection_amd"))
import numpy as np
from zkast import lib
ctx = lib.get_context(0)
rng = np.random.default_rng(9)
W = 2
x = rng.normal(0, 1.0, (W * 1212, 256)).astype(np.float32)
w = rng.normal(0, 0.05, (768, 256)).astype(np.float32)
bias = rng.normal(0, 0.05, (768,)).astype(np.float32)
pos = rng.normal(0, 0.05, (1214, 768)).astype(np.float32)
ref0 = x.astype(np.float64) @ w.astype(np.float64).T + bias
for rep in range(3):
    hid = np.full((W * 1214, 768), 7.0, np.float32)
    out = ctx.test_gemm(x, w, bias, lib.EPI_PATCH, 2, resid=hid, pos=pos).reshape(W, 1214, 768)
    ref = ref0.reshape(W, 1212, 768) + pos[2:]
    d = np.abs(out[:, 2:] - ref); d[np.isnan(d)] = 1e30
    bad = np.argwhere(d > 1e-3)
    print("PATCH rep", rep, "bad count", len(bad), "first", bad[:8].tolist(), "last", bad[-8:].tolist(), flush=True)
    if len(bad):
        ms = sorted(set((b * 1212 + p) for b, p, n in bad.tolist())); ns = sorted(set(n for _, _, n in bad.tolist()))
        print("  bad rows m:", ms[:20], "... n:", ns[:40], flush=True)
for (M, N, K) in [(2424, 768, 256), (2424, 768, 768), (2500, 768, 256)]:
    xx = rng.normal(0, 1.0, (M, K)).astype(np.float32); ww = rng.normal(0, 0.05, (N, K)).astype(np.float32)
    r = xx.astype(np.float64) @ ww.astype(np.float64).T + bias
    o = ctx.test_gemm(xx, ww, bias, lib.EPI_STORE, 2)
    d = np.abs(o - r); d[np.isnan(d)] = 1e30
    bad = np.argwhere(d > 1e-3)
    print("STORE", M, N, K, "bad", len(bad), bad[:6].tolist(), bad[-6:].tolist(), flush=True)
    r0 = np.zeros((M, N), np.float32)
    o = ctx.test_gemm(xx, ww, bias, lib.EPI_RESID, 2, resid=r0)
    d = np.abs(o - r); d[np.isnan(d)] = 1e30
    bad = np.argwhere(d > 1e-3)
    print("RESID", M, N, K, "bad", len(bad), bad[:6].tolist(), bad[-6:].tolist(), flush=True)

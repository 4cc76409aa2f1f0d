import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "zenker-audio-detection_amd")); sys.path.insert(0, ROOT)
import numpy as np
from zkast import lib
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_kernels_gpu import _attn64
ctx = lib.get_context(0)
rng = np.random.default_rng(3)
W = 2
for sc in (1.4, 0.7, 2.0):
    qkv = rng.normal(0, sc, (W * 1214, 2304)).astype(np.float32)
    qkv[:, 1536:] += np.linspace(-1, 1, 768, dtype=np.float32)
    ref = _attn64(qkv, W)
    for ns in (1, 2, 3):
        out = ctx.test_attention(qkv, W, ns)
        d = np.abs(out - ref)
        print(f"scale {sc} nsplit {ns}: max err {d.max():.3e} rms {np.sqrt((d**2).mean()):.3e} (ref max {np.abs(ref).max():.2f})", flush=True)

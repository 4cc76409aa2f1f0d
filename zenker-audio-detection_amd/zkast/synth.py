"""Deterministic synthetic weights and audio for the AST two-stage path.

No checkpoint of the reference's fine-tuned models exists offline, so tests, fixtures and
``bench.py`` all draw the 86,190,338 parameters of ``ASTForAudioClassification(ASTConfig(num_labels=2))``
from a self-contained counter-based PRNG (splitmix64).  Pure integer numpy => bit-identical on every host,
which is what lets the golden logits in ``tests/golden/`` be regenerated without shipping 345 MB of weights.

Tensor names follow the transformers 5.x state-dict layout
(``$TF/models/audio_spectrogram_transformer/modeling_audio_spectrogram_transformer.py:38-318``).
"""
from __future__ import annotations

import numpy as np

HIDDEN = 768
LAYERS = 12
HEADS = 12
INTER = 3072
PATCH = 16
FSTRIDE = 10
TSTRIDE = 10
N_MEL = 128
MAX_LEN = 1024
F_OUT = (N_MEL - PATCH) // FSTRIDE + 1      # 12
T_OUT = (MAX_LEN - PATCH) // TSTRIDE + 1    # 101
SEQ = F_OUT * T_OUT + 2                     # 1214
NUM_LABELS = 2

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def _fnv1a64(name: str) -> int:
    h = 0xCBF29CE484222325
    for b in name.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def splitmix64_uniform(seed: int, n: int) -> np.ndarray:
    """n doubles in [0,1): element i is mix(seed + (i+1)*golden) >> 11, scaled by 2^-53."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, n + 1, dtype=np.uint64)
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + idx * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def _tensor(seed: int, name: str, shape, scale: float, offset: float = 0.0) -> np.ndarray:
    n = int(np.prod(shape))
    u = splitmix64_uniform(seed ^ _fnv1a64(name), n)
    # uniform in [-a, a] with a = scale*sqrt(3) has standard deviation `scale`
    v = (u * 2.0 - 1.0) * (scale * 1.7320508075688772) + offset
    return v.astype(np.float32).reshape(shape)


def ast_param_shapes(num_labels: int = NUM_LABELS) -> dict:
    p = "audio_spectrogram_transformer."
    s = {
        p + "embeddings.cls_token": (1, 1, HIDDEN),
        p + "embeddings.distillation_token": (1, 1, HIDDEN),
        p + "embeddings.position_embeddings": (1, SEQ, HIDDEN),
        p + "embeddings.patch_embeddings.projection.weight": (HIDDEN, 1, PATCH, PATCH),
        p + "embeddings.patch_embeddings.projection.bias": (HIDDEN,),
    }
    for i in range(LAYERS):
        q = f"{p}layers.{i}."
        for nm in ("q_proj", "k_proj", "v_proj", "o_proj"):
            s[q + f"attention.{nm}.weight"] = (HIDDEN, HIDDEN)
            s[q + f"attention.{nm}.bias"] = (HIDDEN,)
        s[q + "layernorm_before.weight"] = (HIDDEN,)
        s[q + "layernorm_before.bias"] = (HIDDEN,)
        s[q + "layernorm_after.weight"] = (HIDDEN,)
        s[q + "layernorm_after.bias"] = (HIDDEN,)
        s[q + "mlp.fc1.weight"] = (INTER, HIDDEN)
        s[q + "mlp.fc1.bias"] = (INTER,)
        s[q + "mlp.fc2.weight"] = (HIDDEN, INTER)
        s[q + "mlp.fc2.bias"] = (HIDDEN,)
    s[p + "layernorm.weight"] = (HIDDEN,)
    s[p + "layernorm.bias"] = (HIDDEN,)
    s["classifier.layernorm.weight"] = (HIDDEN,)
    s["classifier.layernorm.bias"] = (HIDDEN,)
    s["classifier.dense.weight"] = (num_labels, HIDDEN)
    s["classifier.dense.bias"] = (num_labels,)
    return s


# Weight "sets": (matrix std, bias std, embedding std, LN gamma jitter, LN beta std, head std)
WEIGHT_SETS = {
    # HF initializer_range, but with cls/dist/pos randomised too (HF zeroes them,
    # modeling_audio_spectrogram_transformer.py:246-253) so that every term is exercised.
    "init": dict(mat=0.02, bias=0.02, emb=0.02, gamma=0.10, beta=0.05, head=0.05, patch=0.02),
    # wider matrices: peaked attention, residual growth, input-sensitive logits ("trained-like" scale)
    "wide": dict(mat=0.05, bias=0.05, emb=0.05, gamma=0.25, beta=0.10, head=0.10, patch=0.05),
    # what a fine-tuned ViT/DeiT checkpoint looks like to a low-precision GEMM, which the uniform sets are not:
    # heavy-tailed matrices (normal x log-normal scale mixture, kurtosis ~11, max|w| ~ 40-60 sigma) with log-normal
    # per-output-row scales, LayerNorm gains with a handful of 4-8x outlier channels, and two "massive activation"
    # residual channels (fc2 biases of +55 / -40 in layers 2 and 3) that every later LayerNorm has to carry
    "heavy": dict(mat=0.03, bias=0.05, emb=0.05, gamma=0.25, beta=0.10, head=0.10, patch=0.05, heavy=True),
    # INPUT-SENSITIVE: the three sets above see their input through a thick layer of input-independent terms (position
    # embeddings, biases), so their logits move by only ~0.3-1 between very different windows and a front-end error
    # would be attenuated before it reaches them.  Here the patch filters dominate the embedding (std 0.5 against 0.02
    # for positions / cls), attention is content-peaked (std 0.07) and the head has 2x the gain: the six golden windows'
    # logits span > 6 and, with the class-1 bias offset (centred for seed 31, the seed of tests/golden/model_sens.npz), their
    # argmax (the stage-1 gate) flips between windows.
    "sens": dict(mat=0.07, bias=0.02, emb=0.02, gamma=0.25, beta=0.05, head=0.20, patch=0.5, head_bias=(0.0, -5.5)),
}
MASSIVE_CHANNELS = ((47, 55.0), (512, -40.0))      # (residual channel, fc2 bias) of the "heavy" set
MASSIVE_LAYERS = (2, 3)


def _normal_pair(seed: int, name: str, n: int):
    """two independent standard normals per element (Box-Muller on splitmix64 uniforms)"""
    u1 = splitmix64_uniform(seed ^ _fnv1a64(name + "#a"), n)
    u2 = splitmix64_uniform(seed ^ _fnv1a64(name + "#b"), n)
    r = np.sqrt(-2.0 * np.log1p(-u1))
    return r * np.cos(2.0 * np.pi * u2), r * np.sin(2.0 * np.pi * u2)


def _heavy_matrix(seed: int, name: str, shape, scale: float) -> np.ndarray:
    n = int(np.prod(shape))
    z, m = _normal_pair(seed, name, n)
    tau = 0.55                                               # element-wise log-normal mixing: E[exp(2 tau m)] = exp(2 tau^2)
    x = (z * np.exp(tau * m - tau * tau)).reshape(shape[0], -1)
    rz, _ = _normal_pair(seed, name + "#row", shape[0])
    x *= np.exp(0.4 * rz - 0.08)[:, None]                    # per-output-row scale, mean-square 1
    return (x * scale).astype(np.float32).reshape(shape)


def _heavy_gamma(seed: int, name: str, shape, jitter: float) -> np.ndarray:
    g = _tensor(seed, name, shape, jitter, 1.0)
    u = splitmix64_uniform(seed ^ _fnv1a64(name + "#out"), 12)
    for j in range(6):                                       # six outlier channels per LayerNorm, gain 4..8
        g[int(u[2 * j] * shape[0]) % shape[0]] = np.float32(4.0 + 4.0 * u[2 * j + 1])
    return g


def make_ast_weights(seed: int, weight_set: str = "wide", num_labels: int = NUM_LABELS, layers=None) -> dict:
    """Full state dict {name: float32 ndarray}.  `layers` (iterable) restricts which encoder layers are built."""
    ws = WEIGHT_SETS[weight_set]
    out = {}
    for name, shape in ast_param_shapes(num_labels).items():
        if layers is not None and ".layers." in name:
            li = int(name.split(".layers.")[1].split(".")[0])
            if li not in layers:
                continue
        heavy = bool(ws.get("heavy"))
        if name.endswith("layernorm.weight") or name.endswith("layernorm_before.weight") or name.endswith(
            "layernorm_after.weight"
        ):
            out[name] = (_heavy_gamma if heavy and ".layers." in name else lambda a, b, c, d: _tensor(a, b, c, d, 1.0))(
                seed, name, shape, ws["gamma"])
        elif "layernorm" in name and name.endswith(".bias"):
            out[name] = _tensor(seed, name, shape, ws["beta"])
        elif name.endswith(".bias"):
            out[name] = _tensor(seed, name, shape, ws["bias"])
            if name == "classifier.dense.bias" and "head_bias" in ws:
                out[name] = out[name] + np.asarray(ws["head_bias"], dtype=np.float32)[: shape[0]]
            if heavy and name.endswith("mlp.fc2.bias") and int(name.split(".layers.")[1].split(".")[0]) in MASSIVE_LAYERS:
                for ch, val in MASSIVE_CHANNELS:
                    out[name][ch] = np.float32(val)
        elif "embeddings.cls_token" in name or "distillation_token" in name or "position_embeddings" in name:
            out[name] = _tensor(seed, name, shape, ws["emb"])
        elif "patch_embeddings.projection.weight" in name:
            out[name] = _tensor(seed, name, shape, ws["patch"])
        elif name.startswith("classifier.dense"):
            out[name] = _tensor(seed, name, shape, ws["head"])

        elif heavy:
            out[name] = _heavy_matrix(seed, name, shape, ws["mat"])
        else:
            out[name] = _tensor(seed, name, shape, ws["mat"])
    return out


# --------------------------------------------------------------------------------------------------
# synthetic audio
# --------------------------------------------------------------------------------------------------
SR = 16000


def golden_windows() -> np.ndarray:
    """The six 1 s windows pinned by the log-mel fixture (tests/golden/fbank.npz)."""
    rng = np.random.default_rng(1234)
    t = np.arange(SR, dtype=np.float64) / SR
    w = np.zeros((6, SR), dtype=np.float32)
    w[0] = rng.normal(0.0, 0.1, SR).astype(np.float32)
    w[1] = (0.5 * np.sin(2 * np.pi * 440.0 * t)).astype(np.float32)
    w[2] = (0.3 * np.sin(2 * np.pi * (100.0 * t + 0.5 * 7000.0 * t * t))).astype(np.float32)  # 100→7100 Hz chirp
    w[3] = 0.0
    w[4, :5000] = rng.normal(0.0, 0.2, 5000).astype(np.float32)
    w[5] = rng.normal(0.0, 1e-4, SR).astype(np.float32)  # sits on the mel floor
    return w


def synth_recording(seed: int, n_samples: int, burst_rate_hz: float = 0.2) -> np.ndarray:
    """N(0, 0.1^2) noise plus 0.3 s chirp bursts (amp 0.5, 150-1200 Hz) at a Poisson rate."""
    rng = np.random.default_rng(seed)
    x = rng.normal(0.0, 0.1, n_samples).astype(np.float32)
    dur = n_samples / SR
    n_bursts = rng.poisson(burst_rate_hz * dur)
    blen = int(0.3 * SR)
    tb = np.arange(blen, dtype=np.float64) / SR
    for _ in range(int(n_bursts)):
        s = int(rng.integers(0, max(1, n_samples - blen)))
        f0 = rng.uniform(150.0, 600.0)
        f1 = rng.uniform(600.0, 1200.0)
        ph = 2 * np.pi * (f0 * tb + 0.5 * (f1 - f0) / 0.3 * tb * tb)
        env = np.hanning(blen)
        seg = (0.5 * env * np.sin(ph)).astype(np.float32)
        e = min(n_samples, s + blen)
        x[s:e] += seg[: e - s]
    return x


def synth_windows(seed: int, n_windows: int) -> np.ndarray:
    """(n_windows, 16000) fp32: 1 s / 0.5 s-hop windows cut from one synthetic recording."""
    n_samples = SR + (n_windows - 1) * (SR // 2)
    x = synth_recording(seed, n_samples)
    idx = np.arange(n_windows)[:, None] * (SR // 2) + np.arange(SR)[None, :]
    return x[idx]

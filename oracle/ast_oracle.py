"""CPU ORACLE — TEST INFRASTRUCTURE ONLY.  Not shipped, not on the product path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module,
and only as the checker.  The product path (``zkast`` -> ``libzkast.so`` -> HIP kernels) never imports it and
fails loudly when the HIP library is missing.

This is a numpy restatement of the arithmetic of the reference's hot path
(``src/test_long_audio_windows_2stage.py:62-113,148-195,312-340``).  That arithmetic lives in the third-party
package the reference calls, ``transformers`` (requirements.txt:4, ``>=4.30.0``; pinned here by the copy in this
image, 5.15.0), cited below as ``$TF/...``:

* log-mel:  ``$TF/models/audio_spectrogram_transformer/feature_extraction_audio_spectrogram_transformer.py:92-158``
  (numpy branch, used when torchaudio is absent) and ``$TF/audio_utils.py:448-560,638-729,745-800,809-1017``.
* AST forward: ``$TF/models/audio_spectrogram_transformer/modeling_audio_spectrogram_transformer.py:38-380``.

PINNING: the reference has no tests or golden vectors of its own.  This restatement is pinned by
fixtures generated in the build container from the real ``transformers`` classes and the reference module's own
pure functions (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz|json``); ``tests/test_oracle.py`` checks
it against every one of them.  The torchaudio-kaldi extractor branch and torchaudio's resampler are *parity
unpinned* (their source is not in the image).
"""
from __future__ import annotations

import numpy as np

try:  # scipy is in the image; only erf is needed
    from scipy.special import erf as _erf
except Exception:  # pragma: no cover
    import math

    _erf = np.vectorize(math.erf)

SR = 16000
N_MEL = 128
MAX_LEN = 1024
FRAME_LEN = 400
HOP_LEN = 160
FFT_LEN = 512
N_BINS = FFT_LEN // 2 + 1
MEL_FLOOR = 1.192092955078125e-07
PREEMPH = 0.97
HIDDEN, HEADS, HEAD_DIM, INTER, LAYERS = 768, 12, 64, 3072, 12
PATCH, FSTRIDE, TSTRIDE = 16, 10, 10
F_OUT, T_OUT = 12, 101
SEQ = F_OUT * T_OUT + 2
LN_EPS = 1e-12


# --------------------------------------------------------------------------------------------------
# window indexing  (src/test_long_audio_windows_2stage.py:62-75)
# --------------------------------------------------------------------------------------------------
def window_starts(n_samples: int, window_sec: float = 1.0, hop_sec: float = 0.5, sr: int = SR):
    win = int(window_sec * sr)
    hop = int(hop_sec * sr)
    return list(range(0, max(1, n_samples - win + 1), hop)), win, hop


def window_audio(audio: np.ndarray, window_sec: float = 1.0, hop_sec: float = 0.5, sr: int = SR):
    starts, win, _ = window_starts(len(audio), window_sec, hop_sec, sr)
    out = []
    for s in starts:
        seg = audio[s : s + win]
        if len(seg) < win:  # only when the whole recording is shorter than one window
            pad = np.zeros(win, dtype=audio.dtype)
            pad[: len(seg)] = seg
            seg = pad
        out.append(seg)
    return out


# --------------------------------------------------------------------------------------------------
# mel filter bank + window  ($TF/audio_utils.py:467-468,503-504,541-560,638-729,778-785)
# --------------------------------------------------------------------------------------------------
def hertz_to_mel_kaldi(f):
    return 1127.0 * np.log(1.0 + (np.asarray(f, dtype=np.float64) / 700.0))


def mel_filter_bank_kaldi() -> np.ndarray:
    """(257,128) float64; triangles built in mel space, norm=None, 20 Hz .. 8000 Hz."""
    mel_min = hertz_to_mel_kaldi(20.0)
    mel_max = hertz_to_mel_kaldi(float(SR // 2))
    mel_freqs = np.linspace(mel_min, mel_max, N_MEL + 2)
    fft_bin_width = SR / ((N_BINS - 1) * 2)
    fft_freqs = hertz_to_mel_kaldi(fft_bin_width * np.arange(N_BINS))
    filter_diff = np.diff(mel_freqs)
    slopes = np.expand_dims(mel_freqs, 0) - np.expand_dims(fft_freqs, 1)
    down = -slopes[:, :-2] / filter_diff[:-1]
    up = slopes[:, 2:] / filter_diff[1:]
    return np.maximum(np.zeros(1), np.minimum(down, up))


def hann_window() -> np.ndarray:
    """symmetric Hann-400 (window_function(400,'hann',periodic=False) == np.hanning(400))."""
    n = np.arange(FRAME_LEN, dtype=np.float64)
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * n / (FRAME_LEN - 1))


# Built once at import and published as ONE immutable tuple: there is no lazily initialised state a worker thread of
# ast_torch_cpu.extract_features_parallel could observe half-built (round 3's bench died on exactly that: `_MEL` was
# assigned before `_HANN`, a second thread saw `_MEL is not None` and got `(mel, None)`).
_CONSTS = (mel_filter_bank_kaldi(), hann_window())
for _a in _CONSTS:
    _a.setflags(write=False)


def _consts():
    return _CONSTS


# --------------------------------------------------------------------------------------------------
# log-mel  ($TF/audio_utils.py:956-1015; feature_extraction…:106-158)
# --------------------------------------------------------------------------------------------------
def fbank_frames(waveform: np.ndarray) -> np.ndarray:
    """(T,) fp32 -> (num_frames,128) fp32 un-normalised log-mel; float64 framing, complex64 spectrum storage."""
    mel, hann = _consts()
    x = np.asarray(waveform, dtype=np.float32).astype(np.float64)
    n_frames = int(1 + np.floor((x.size - FRAME_LEN) / HOP_LEN))
    idx = np.arange(n_frames)[:, None] * HOP_LEN + np.arange(FRAME_LEN)[None, :]
    fr = x[idx]                                           # (F,400)
    fr = fr - fr.mean(axis=1, keepdims=True)              # remove_dc_offset (:979-980)
    pe = fr.copy()
    pe[:, 1:] = fr[:, 1:] - PREEMPH * fr[:, :-1]          # (:982-984)
    pe[:, 0] = fr[:, 0] * (1.0 - PREEMPH)
    pe *= hann[None, :]                                   # (:986)
    buf = np.zeros((n_frames, FFT_LEN), dtype=np.float64)
    buf[:, :FRAME_LEN] = pe
    spec = np.fft.rfft(buf, axis=1).astype(np.complex64)  # stored as complex64 (:966,:988)
    power = np.abs(spec, dtype=np.float64) ** 2.0         # (:993)
    melspec = np.maximum(MEL_FLOOR, power @ mel)          # (:998)   (F,128)
    return np.log(melspec).astype(np.float32)             # (:1002,:1015)


def fbank_frames_kaldi_fp32(waveform: np.ndarray) -> np.ndarray:
    """PARITY UNPINNED.  (T,) fp32 -> (num_frames,128) fp32: the OTHER branch of ``_extract_fbank_features``
    ($TF/…/feature_extraction_audio_spectrogram_transformer.py:116-123), taken when torchaudio is installed — which a
    reference install has (requirements.txt:12; src/test_long_audio_windows_2stage.py:39 imports it unconditionally):
    ``torchaudio.compliance.kaldi.fbank(waveform, sample_frequency=16000, window_type="hanning", num_mel_bins=128)``.

    torchaudio is NOT in this image, so this is a restatement of its published algorithm from the library's
    documentation and general knowledge of ``torchaudio/compliance/kaldi.py`` (defaults: 25 ms / 10 ms frames, snip_edges,
    dither 0, remove_dc_offset, pre-emphasis 0.97 over a replicate-padded frame, symmetric Hann, zero-pad to 512,
    ``torch.fft.rfft`` -> |.|^2, mel banks built in FLOAT32 from ``1127 ln(1 + f/700)`` with the Nyquist bin's weight
    forced to zero, ``max(eps_fp32).log()``), written with the same torch float32 operators torchaudio uses.  No fixture
    can pin it here; it exists to MEASURE how far the two branches move the features and the logits
    (tests/test_oracle.py::test_extractor_branch_gap).  Same arithmetic as the numpy branch, but float32 throughout
    instead of float64 framing + complex64 spectrum."""
    import torch
    x = torch.from_numpy(np.ascontiguousarray(waveform, dtype=np.float32))
    n_frames = 1 + (x.numel() - FRAME_LEN) // HOP_LEN
    fr = x.as_strided((n_frames, FRAME_LEN), (HOP_LEN, 1))                       # _get_strided, snip_edges=True
    fr = fr - fr.mean(dim=1, keepdim=True)                                       # remove_dc_offset
    prev = torch.nn.functional.pad(fr.unsqueeze(0), (1, 0), mode="replicate").squeeze(0)[:, :-1]
    fr = fr - PREEMPH * prev                                                     # x[0] - 0.97 x[0] at the left edge
    fr = fr * torch.hann_window(FRAME_LEN, periodic=False, dtype=torch.float32)  # window_type="hanning"
    fr = torch.nn.functional.pad(fr, (0, FFT_LEN - FRAME_LEN))                   # round_to_power_of_two
    power = torch.fft.rfft(fr).abs().pow(2.0)                                    # use_power=True, (F,257) fp32
    # get_mel_banks(128, 512, 16000, low 20, high 0 -> nyquist), all float32
    mel = lambda f: 1127.0 * torch.log(1.0 + f / 700.0)                          # noqa: E731
    lo, hi = mel(torch.tensor(20.0)), mel(torch.tensor(float(SR // 2)))
    delta = (hi - lo) / (N_MEL + 1)
    b = torch.arange(N_MEL, dtype=torch.float32).unsqueeze(1)
    left, center, right = lo + b * delta, lo + (b + 1.0) * delta, lo + (b + 2.0) * delta
    fft_mel = mel((SR / FFT_LEN) * torch.arange(FFT_LEN // 2, dtype=torch.float32)).unsqueeze(0)      # 256 bins
    banks = torch.clamp(torch.minimum((fft_mel - left) / (center - left), (right - fft_mel) / (right - center)), min=0.0)
    banks = torch.nn.functional.pad(banks, (0, 1))                               # Nyquist column = 0 -> (128,257)
    e = torch.mm(power, banks.T)
    return torch.clamp(e, min=float(np.finfo(np.float32).eps)).log().numpy()     # use_log_fbank


def extract_features(windows, mean: float, std: float, do_normalize: bool = True, branch: str = "numpy") -> np.ndarray:
    """list/array of (16000,) fp32 -> (B,1024,128) fp32, i.e. ASTFeatureExtractor.__call__(...)['input_values'].
    branch "numpy" (pinned, the branch this image and the build use) or "kaldi" (parity unpinned, see above)."""
    frames = fbank_frames if branch == "numpy" else fbank_frames_kaldi_fp32
    out = np.zeros((len(windows), MAX_LEN, N_MEL), dtype=np.float32)
    for i, w in enumerate(windows):
        fb = frames(np.squeeze(np.asarray(w, dtype=np.float32)))
        n = min(fb.shape[0], MAX_LEN)
        out[i, :n] = fb[:n]                               # ZeroPad2d / truncate (:143-151)
    if do_normalize:
        # (x - mean) / (std*2) on float32 arrays with python-float scalars (:157-158)
        out = ((out - np.float32(mean)) / np.float32(std * 2)).astype(np.float32)
    return out


# --------------------------------------------------------------------------------------------------
# quantisation emulation (what an f16/bf16-input, fp32-accumulate MFMA sees)
# --------------------------------------------------------------------------------------------------
def _q(x: np.ndarray, mode) -> np.ndarray:
    if mode is None:
        return x
    x = np.asarray(x, dtype=np.float32)
    if mode == "f16":
        return x.astype(np.float16).astype(np.float32)
    if mode == "bf16":
        u = x.view(np.uint32).astype(np.uint64)
        r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
        return (r & 0xFFFFFFFF).astype(np.uint32).view(np.float32)
    raise ValueError(mode)


# --------------------------------------------------------------------------------------------------
# AST forward  ($TF/…/modeling_audio_spectrogram_transformer.py)
# --------------------------------------------------------------------------------------------------
def _ln(x, g, b):
    x64 = x.astype(np.float64)
    mu = x64.mean(-1, keepdims=True)
    var = ((x64 - mu) ** 2).mean(-1, keepdims=True)
    return (((x64 - mu) / np.sqrt(var + LN_EPS)) * g + b).astype(np.float32)


def _gelu(x):
    return (0.5 * x * (1.0 + _erf(x / np.sqrt(2.0)))).astype(np.float32)


def _lin(x, w, b, quant):
    if quant == "f16c8":
        return _lin_f16c8(x, w, b)
    return _q(x, quant) @ _q(w, quant).T + b


def _lin_f16c8(x, w, b):
    """What the ZK_F16C8 GEMM computes (gemm_c8.hip): fp16 product + the two split corrections in e4m3,
    x·w = xh·wh + [fp8(xl·2^11)·fp8(w·2^e) + fp8(x)·fp8(wl·2^(e+11))]·2^-(e+11), fp32 accumulation."""
    x = np.asarray(x, np.float32); w = np.asarray(w, np.float32)
    xh = x.astype(np.float16).astype(np.float32); wh = w.astype(np.float16).astype(np.float32)
    wmax = float(np.abs(w).max())
    e = int(np.floor(np.log2(224.0 / wmax))) if wmax > 0 else 0
    x8 = fp8_e4m3_round(x); xl8 = fp8_e4m3_round((x - xh) * np.float32(2048.0))
    w8 = fp8_e4m3_round(w * np.float32(2.0 ** e)); wl8 = fp8_e4m3_round((w - wh) * np.float32(2.0 ** (e + 11)))
    corr = (xl8 @ w8.T + x8 @ wl8.T) * np.float32(2.0 ** -(e + 11))
    return xh @ wh.T + corr + b


class ASTWeights:
    """Accepts the transformers 5.x or 4.x key scheme ($TF/conversion_mapping.py:134,338-346)."""

    def __init__(self, sd: dict):
        self.sd = {k: np.asarray(v, dtype=np.float32) for k, v in sd.items()}
        self.p = "audio_spectrogram_transformer."
        self.v5 = any(".layers." in k and "encoder.layer." not in k for k in self.sd)

    def g(self, name):
        return self.sd[name]

    def layer(self, i):
        p = self.p
        if self.v5:
            q = f"{p}layers.{i}."
            names = dict(q="attention.q_proj", k="attention.k_proj", v="attention.v_proj", o="attention.o_proj",
                         fc1="mlp.fc1", fc2="mlp.fc2", ln1="layernorm_before", ln2="layernorm_after")
        else:
            q = f"{p}encoder.layer.{i}."
            names = dict(q="attention.attention.query", k="attention.attention.key", v="attention.attention.value",
                         o="attention.output.dense", fc1="intermediate.dense", fc2="output.dense",
                         ln1="layernorm_before", ln2="layernorm_after")
        return {k: (self.sd[q + v + ".weight"], self.sd[q + v + ".bias"]) for k, v in names.items()}

    @property
    def n_layers(self):
        n = 0
        while True:
            key = (f"{self.p}layers.{n}.layernorm_before.weight" if self.v5
                   else f"{self.p}encoder.layer.{n}.layernorm_before.weight")
            if key not in self.sd:
                return n
            n += 1


def embed(input_values: np.ndarray, W: ASTWeights, quant=None) -> np.ndarray:
    """(B,1024,128) -> (B,1214,768).  Conv2d(1,768,16x16,stride 10) over [freq,time] as an im2col GEMM (:57-61,:89-99)."""
    B = input_values.shape[0]
    p = W.p + "embeddings."
    cw = W.g(p + "patch_embeddings.projection.weight").reshape(HIDDEN, PATCH * PATCH)  # [out, kf*16+kt]
    cb = W.g(p + "patch_embeddings.projection.bias")
    x = input_values.transpose(0, 2, 1)                   # (B, freq 128, time 1024)
    fi = (np.arange(F_OUT) * FSTRIDE)[:, None] + np.arange(PATCH)[None, :]   # (12,16)
    ti = (np.arange(T_OUT) * TSTRIDE)[:, None] + np.arange(PATCH)[None, :]   # (101,16)
    patches = x[:, fi[:, None, :, None], ti[None, :, None, :]]               # (B,12,101,16,16)
    patches = patches.reshape(B, F_OUT * T_OUT, PATCH * PATCH)
    emb = _lin(patches, cw, cb, quant)                     # token = f*101 + t
    cls = np.broadcast_to(W.g(p + "cls_token"), (B, 1, HIDDEN))
    dist = np.broadcast_to(W.g(p + "distillation_token"), (B, 1, HIDDEN))
    h = np.concatenate([cls, dist, emb], axis=1) + W.g(p + "position_embeddings")
    return h.astype(np.float32)


def encoder_layer(h: np.ndarray, L: dict, quant=None) -> np.ndarray:
    B, S, _ = h.shape
    x = _ln(h, *L["ln1"])
    q = _lin(x, *L["q"], quant).reshape(B, S, HEADS, HEAD_DIM).transpose(0, 2, 1, 3)
    k = _lin(x, *L["k"], quant).reshape(B, S, HEADS, HEAD_DIM).transpose(0, 2, 1, 3)
    v = _lin(x, *L["v"], quant).reshape(B, S, HEADS, HEAD_DIM).transpose(0, 2, 1, 3)
    aq = None if quant == "f16c8" else quant        # f16c8: QK^T is a 3-term fp16 split (fp32-grade), P.V one fp16 pass
    s = (_q(q, aq) @ _q(k, aq).transpose(0, 1, 3, 2)) * np.float32(HEAD_DIM ** -0.5)
    s = s - s.max(-1, keepdims=True)
    e = np.exp(s)
    pq = "f16" if quant == "f16c8" else quant
    if pq is None:
        a = (e / e.sum(-1, keepdims=True)) @ v
    else:       # what attention.hip computes: un-normalised weights rounded for the MFMA, row sum over the ROUNDED weights;
        eq = _q(e, pq)      # v enters as (hi, lo) fp16 pair (P·Vh + P·Vl), i.e. to ~2^-22: unrounded here
        a = (eq @ v) / eq.sum(-1, keepdims=True)
    a = a.transpose(0, 2, 1, 3).reshape(B, S, HIDDEN)
    h = h + _lin(a, *L["o"], quant)
    x = _ln(h, *L["ln2"])
    m = _gelu(_lin(x, *L["fc1"], quant))
    return (h + _lin(m, *L["fc2"], quant)).astype(np.float32)


def ast_forward(input_values: np.ndarray, sd, quant=None, return_hidden: bool = False, chunk: int = 4):
    """(B,1024,128) fp32 -> logits (B,num_labels) fp32 [, dict of checkpoints]."""
    W = sd if isinstance(sd, ASTWeights) else ASTWeights(sd)
    outs, hid = [], {}
    nl = W.n_layers
    layers = [W.layer(i) for i in range(nl)]
    for b0 in range(0, input_values.shape[0], chunk):
        h = embed(input_values[b0 : b0 + chunk], W, quant)
        ck = {"emb": h}
        for i in range(nl):
            h = encoder_layer(h, layers[i], quant)
            if return_hidden:
                ck[f"layer{i}"] = h
        seq = _ln(h, W.g(W.p + "layernorm.weight"), W.g(W.p + "layernorm.bias"))
        pooled = (seq[:, 0] + seq[:, 1]) / 2                                  # (:304)
        z = _ln(pooled, W.g("classifier.layernorm.weight"), W.g("classifier.layernorm.bias"))
        logits = z @ W.g("classifier.dense.weight").T + W.g("classifier.dense.bias")
        outs.append(logits.astype(np.float32))
        if return_hidden:
            ck["final_ln"] = seq
            ck["pooled"] = pooled
            for k_, v_ in ck.items():
                hid.setdefault(k_, []).append(v_)
    logits = np.concatenate(outs, 0) if outs else np.zeros((0, 2), np.float32)
    if return_hidden:
        return logits, {k_: np.concatenate(v_, 0) for k_, v_ in hid.items()}
    return logits


def softmax(logits: np.ndarray) -> np.ndarray:
    z = logits - logits.max(axis=1, keepdims=True)
    e = np.exp(z)
    return (e / e.sum(axis=1, keepdims=True)).astype(np.float32)


def forward_probs(sd, mean: float, std: float, windows, batch_size: int = 128, quant=None) -> np.ndarray:
    """src/test_long_audio_windows_2stage.py:104-113 with the oracle extractor + model."""
    W = sd if isinstance(sd, ASTWeights) else ASTWeights(sd)
    out = []
    for i in range(0, len(windows), batch_size):
        feats = extract_features(windows[i : i + batch_size], mean, std)
        out.append(softmax(ast_forward(feats, W, quant)))
    return np.concatenate(out, 0) if out else np.zeros((0,))


# --------------------------------------------------------------------------------------------------
# cascade decisions + summary  (src/test_long_audio_windows_2stage.py:148-195,312-340)
# --------------------------------------------------------------------------------------------------
def stage1_gate(s1_probs: np.ndarray, thr1: float, fwd_min_prob=None) -> np.ndarray:
    p_sw = s1_probs[:, 1]
    pred = s1_probs.argmax(axis=1)
    pred = np.where((pred == 1) & (p_sw >= thr1), 1, 0)
    idx = np.where(pred == 1)[0]
    if fwd_min_prob is not None:  # cache variant, …_cache.py:471-478
        idx = idx[p_sw[idx] >= fwd_min_prob]
    return idx


def summarize_stage_outputs(stage1_probs, stage2_results, stage2_threshold: float = 0.5, use_argmax: bool = False):
    s1_preds = stage1_probs.argmax(axis=1)
    aligned = [None] * len(s1_preds)
    for idx, pr in stage2_results:
        aligned[idx] = pr
    idle = int((s1_preds == 0).sum())
    swallow = int((s1_preds == 1).sum())
    ev = [p for p in aligned if p is not None]
    if use_argmax:  # …_cache.py:258-265
        healthy = int(sum(1 for p in ev if int(np.argmax(p)) == 0))
        zenker = int(sum(1 for p in ev if int(np.argmax(p)) == 1))
    else:
        healthy = int(sum(1 for p in ev if p[1] < stage2_threshold))
        zenker = int(sum(1 for p in ev if p[1] >= stage2_threshold))
    n = len(s1_preds)
    return {
        "num_windows": int(n),
        "stage1_idle_windows": idle,
        "stage1_swallow_windows": swallow,
        "stage1_swallow_ratio": (swallow / n) if n else 0.0,
        "stage1_mean_probs": stage1_probs.mean(axis=0).tolist() if len(stage1_probs) else None,
        "stage2_mean_probs_over_swallow": np.mean(ev, axis=0).tolist() if swallow else None,
        "stage2_swallow_windows_evaluated": int(len(ev)),
        "stage2_healthy_windows": healthy,
        "stage2_zenker_windows": zenker,
        "stage2_zenker_ratio_over_swallow": (zenker / swallow) if swallow else None,
    }


# --------------------------------------------------------------------------------------------------
# load_audio's resampler (src/test_long_audio_windows_2stage.py:57-58) — PARITY UNPINNED
# --------------------------------------------------------------------------------------------------
def resample_sinc_hann(x: np.ndarray, orig_sr: int, new_sr: int, lowpass_filter_width: int = 6,
                       rolloff: float = 0.99) -> np.ndarray:
    """torchaudio.functional.resample defaults, restated from the published algorithm
    (torchaudio/functional/functional.py `_get_sinc_resample_kernel` / `_apply_sinc_resample_kernel`, v2.x).
    torchaudio is not installed in the build image and the reference pins only `>=2.0.0` (requirements.txt:12),
    so nothing here is checked against the real library: parity unpinned."""
    g = int(np.gcd(int(orig_sr), int(new_sr)))
    orig, new = int(orig_sr) // g, int(new_sr) // g
    if orig == new:
        return np.asarray(x, dtype=np.float32)
    base_freq = min(orig, new) * rolloff
    width = int(np.ceil(lowpass_filter_width * orig / base_freq))
    idx = np.arange(-width, width + orig, dtype=np.float64)[None, :] / orig
    t = np.arange(0, -new, -1, dtype=np.float64)[:, None] / new + idx
    t = np.clip(t * base_freq, -lowpass_filter_width, lowpass_filter_width)
    window = np.cos(t * np.pi / lowpass_filter_width / 2) ** 2
    t = t * np.pi
    scale = base_freq / orig
    kern = np.where(t == 0, 1.0, np.sin(t) / np.where(t == 0, 1.0, t)) * window * scale   # (new, 2w+orig)
    kern = kern.astype(np.float32)
    x = np.asarray(x, dtype=np.float32)
    n = x.shape[0]
    padded = np.concatenate([np.zeros(width, np.float32), x, np.zeros(width + orig, np.float32)])
    n_frames = (padded.shape[0] - kern.shape[1]) // orig + 1
    frames = np.lib.stride_tricks.sliding_window_view(padded, kern.shape[1])[::orig][:n_frames]
    out = (frames.astype(np.float64) @ kern.T.astype(np.float64)).reshape(-1)        # (frames*new,)
    target = int(np.ceil(new * n / orig))
    return out[:target].astype(np.float32)


# --------------------------------------------------------------------------------------------------
# c8 operand planes of the ZK_F16C8 GEMM mode (checker for zk_test_split_c8; format: include/zkast.h)
# --------------------------------------------------------------------------------------------------
def fp8_e4m3_bits(x: np.ndarray) -> np.ndarray:
    """float32 -> OCP e4m3 (bias 7, no infinities, max 448) bit patterns, round to nearest even, inputs clamped to
    +-448.  Pure integer/floor arithmetic (no float8 dtype needed)."""
    x = np.clip(np.asarray(x, dtype=np.float64), -448.0, 448.0)
    sign = (np.signbit(x)).astype(np.uint8) << 7
    a = np.abs(x)
    # exponent of the value's binade, clamped to the subnormal binade (2^-6): step = 2^(e-3)
    with np.errstate(divide="ignore"):
        e = np.floor(np.log2(np.where(a > 0, a, 2.0 ** -12)))      # zero lands in the subnormal binade
    e = np.maximum(e, -6.0)
    step = np.exp2(e - 3.0)
    q = np.rint(a / step)                         # numpy rint = round half to even
    # q in [0, 16]: 16 means carry into the next binade
    carry = q >= 16
    e = np.where(carry, e + 1, e)
    q = np.where(carry, 8.0, q)
    is_sub = (e == -6) & (q < 8)
    exp_field = np.where(is_sub, 0, e + 7).astype(np.int64)
    man = np.where(is_sub, q, q - 8).astype(np.int64)
    return (sign | (exp_field.astype(np.uint8) << 3) | man.astype(np.uint8)).astype(np.uint8)


def _e4m3_lut() -> np.ndarray:
    b = np.arange(256)
    s_, e_, m_ = b >> 7, (b >> 3) & 15, b & 7
    v = np.where(e_ == 0, m_ * 2.0 ** -9, (8 + m_) * np.exp2(e_ - 10.0))
    return np.where(s_ == 1, -v, v).astype(np.float32)


_E4M3_LUT = _e4m3_lut()          # import-time constant, like _CONSTS: no lazily built module state anywhere in the oracle


def fp8_e4m3_round(x: np.ndarray) -> np.ndarray:
    """x rounded to the nearest e4m3 value (clamped to +-448), as float32."""
    return _E4M3_LUT[fp8_e4m3_bits(x)]


def c8_plane(x: np.ndarray, w_exp: int = 0, is_weight: bool = False) -> np.ndarray:
    """uint16 entries of the c8 plane: activations (lo8, x8), weights (w8, lo8) — byte 0 is the low byte."""
    x = np.asarray(x, dtype=np.float32)
    lo = (x - x.astype(np.float16).astype(np.float32)).astype(np.float32)
    if is_weight:
        b0 = fp8_e4m3_bits(x * np.float32(2.0 ** w_exp))
        b1 = fp8_e4m3_bits(lo * np.float32(2.0 ** (w_exp + 11)))
    else:
        b0 = fp8_e4m3_bits(lo * np.float32(2048.0))
        b1 = fp8_e4m3_bits(x)
    return b0.astype(np.uint16) | (b1.astype(np.uint16) << 8)

"""Host-side mirror of the reference's two-stage long-audio script, function for function
(src/test_long_audio_windows_2stage.py and the options the cached variant adds,
src/test_long_audio_windows_2stage_cache.py): same names, argument meaning, return types and JSON schema, with the
feature extraction, both AST forwards, the softmax and the gate running as HIP kernels behind libzkast.so.
"""
from __future__ import annotations

import argparse
import fnmatch
import json
import os
import struct
from typing import Any, Dict, List, Optional, Tuple

import numpy as np

from . import lib as _lib
from .lib import DEFAULT_COMPUTE_MODE
from .feature_extraction import ZkASTFeatureExtractor
from .modeling import ZkASTConfig, ZkASTForAudioClassification

SAMPLING_RATE = 16000


# ----------------- Audio helpers -----------------
def parse_wav(path: str):
    """RIFF/WAVE container walk -> (format_tag, channels, sample_rate, bits, sample_bytes).  PCM 8/16/24/32-bit, IEEE float
    32/64, WAVE_FORMAT_EXTENSIBLE."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file")
    pos, fmt, raw = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            tag, ch, sr, _br, _ba, bits = struct.unpack("<HHIIHH", body[:16])
            if tag == 0xFFFE and len(body) >= 26:
                tag = struct.unpack("<H", body[24:26])[0]
            fmt = (tag, ch, sr, bits)
        elif cid == b"data":
            raw = body
        pos += 8 + size + (size & 1)
    if fmt is None or raw is None:
        raise ValueError(f"{path}: missing fmt/data chunk")
    tag, ch, sr, bits = fmt
    if not ((tag == 1 and bits in (8, 16, 24, 32)) or (tag == 3 and bits in (32, 64))):
        raise ValueError(f"{path}: unsupported WAVE format tag {tag} / {bits} bits")
    return tag, ch, sr, bits, raw


def read_wav(path: str) -> Tuple[np.ndarray, int]:
    """Minimal RIFF/WAVE reader (host, numpy) -> (channels, frames) float32 in [-1, 1], sample rate.  Stands in for
    torchaudio.load (src/test_long_audio_windows_2stage.py:54); the files it reads are written by
    utils/PrepareDatasetLongAudio.py:59-67 (soundfile, mono PCM_16, native rate).  The product path decodes on the GPU
    (load_audio -> zk_wav_decode); this host version is what the dataset-preparation helper and the tests use."""
    tag, ch, sr, bits, raw = parse_wav(path)
    if tag == 1:
        if bits == 16:
            x = np.frombuffer(raw[: len(raw) // 2 * 2], "<i2").astype(np.float32) / 32768.0
        elif bits == 8:
            x = (np.frombuffer(raw, np.uint8).astype(np.float32) - 128.0) / 128.0
        elif bits == 24:
            b = np.frombuffer(raw[: len(raw) // 3 * 3], np.uint8).reshape(-1, 3).astype(np.int32)
            v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
            v = np.where(v >= 1 << 23, v - (1 << 24), v)
            x = v.astype(np.float32) / float(1 << 23)
        else:
            x = (np.frombuffer(raw[: len(raw) // 4 * 4], "<i4").astype(np.float64) / float(1 << 31)).astype(np.float32)
    else:
        x = np.frombuffer(raw[: len(raw) // (bits // 8) * (bits // 8)], "<f4" if bits == 32 else "<f8").astype(np.float32)
    n = x.shape[0] // ch
    return np.ascontiguousarray(x[: n * ch].reshape(n, ch).T), sr


def write_wav_pcm16(path: str, audio: np.ndarray, sr: int):
    a = np.clip(np.asarray(audio, dtype=np.float64), -1.0, 1.0 - 1.0 / 32768)
    pcm = np.round(a * 32768.0).astype("<i2").tobytes()
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(pcm)) + b"WAVE" + b"fmt " +
                struct.pack("<IHHIIHH", 16, 1, 1, sr, sr * 2, 2, 16) + b"data" + struct.pack("<I", len(pcm)) + pcm)


def load_audio_to_device(path: str, target_sr: int = SAMPLING_RATE, device: int = 0) -> int:
    """load_audio (src/test_long_audio_windows_2stage.py:53-59) that STAYS on the GPU: the RIFF header is walked here,
    the data chunk goes up once, sample decode + channel mean + resampling run on the device (zk_audio_load) and the
    recording is left in the context's audio slot for classify_recording(audio=None).  Returns its length."""
    tag, ch, sr, bits, raw = parse_wav(path)
    return _lib.get_context(device).audio_load(raw, tag, bits, ch, sr, target_sr)


def load_audio(path: str, target_sr: int = SAMPLING_RATE, device: int = 0) -> np.ndarray:
    """:53-59 with the reference's return contract (1-D np.float32 at 16 kHz): load_audio_to_device + one copy back."""
    load_audio_to_device(path, target_sr, device)
    return _lib.get_context(device).audio_get()


def window_geometry(n_samples: int, window_sec: float, hop_sec: float, sr: int = SAMPLING_RATE):
    """(n_windows, win, hop) of window_audio (:62-75): starts range(0, max(1, T-win+1), hop)."""
    win = int(window_sec * sr)
    hop = int(hop_sec * sr)
    if win <= 0 or hop <= 0:
        raise ValueError("window-sec and hop-sec must be > 0")
    return len(range(0, max(1, n_samples - win + 1), hop)), win, hop


def window_audio(audio: np.ndarray, window_sec: float, hop_sec: float, sr: int = SAMPLING_RATE) -> List[np.ndarray]:
    """The window list of the reference's window_audio (:62-75) for callers that want it materialised (the product path
    never does: the kernels cut windows out of the recording by index).  Full windows are zero-copy slices of `audio`
    (they alias the recording and, at hop < window, each other — exactly like the reference's); a recording shorter than
    one window gives its single zero-padded window."""
    n, win, _hop = window_geometry(len(audio), window_sec, hop_sec, sr)
    if len(audio) < win:
        only = np.zeros(win, dtype=audio.dtype)
        only[: len(audio)] = audio
        return [only]
    return [audio[i * _hop: i * _hop + win] for i in range(n)]      # plain slices, as the reference's: writable iff `audio` is


# ----------------- Model loading -----------------
def load_stage_model(model_root: str, label_order: List[str], stage: int = 0, compute_mode=DEFAULT_COMPUTE_MODE, device: int = 0):
    """(fx, model) of one stage from a local checkpoint directory, labels set from `label_order` (:86-98).  `stage`
    picks the library's weight slot (0: Idle/Swallow, 1: Healthy/Zenker); both stages stay resident side by side."""
    config = ZkASTConfig.from_pretrained(model_root)
    config.id2label = dict(enumerate(label_order))
    config.label2id = {name: i for i, name in config.id2label.items()}
    fx = ZkASTFeatureExtractor.from_pretrained(model_root, device=device)
    model = ZkASTForAudioClassification.from_pretrained(model_root, config=config, stage=stage,
                                                        compute_mode=compute_mode, device=device)
    return fx, model.eval().bind_feature_extractor(fx)


# ----------------- Inference -----------------
def forward_probs(model, fx, windows: List[np.ndarray], batch_size: int) -> np.ndarray:
    """Contract of the reference's forward_probs (:104-113): a list of 1-D float32 windows -> (N, labels) float32
    softmax probabilities, np.zeros((0,)) for an empty list — through the drop-in extractor and model objects, one
    `batch_size` slice at a time, so a script written against the HuggingFace classes runs unchanged."""
    total = len(windows)
    if total == 0:
        return np.zeros((0,))
    ctx = _lib.get_context(getattr(model, "_device", 0))
    key = fx.model_input_names[0]
    probs = np.empty((total, model.num_labels), dtype=np.float32)
    for lo in range(0, total, batch_size):
        hi = min(total, lo + batch_size)
        feats = fx(windows[lo:hi], sampling_rate=SAMPLING_RATE, return_tensors="np")[key]
        probs[lo:hi] = ctx.softmax(np.asarray(model(feats).logits))
    return probs


def forward_probs_recording(model, fx, audio: np.ndarray, window_sec: float, hop_sec: float,
                            win_idx=None) -> np.ndarray:
    """Fused equivalent of forward_probs(model, fx, window_audio(audio, ...)): the windows are never materialised,
    the log-mel stays on the device, and the (B,1024,128) tensor is never built."""
    n, win, hop = window_geometry(len(audio), window_sec, hop_sec)
    ctx = model._ctx
    model.bind_feature_extractor(fx)
    ctx.logmel(np.ascontiguousarray(audio, dtype=np.float32), len(audio), 0, hop, win, n)
    logits = model.forward_from_slot(n, win_idx)
    return ctx.softmax(logits) if logits.shape[0] else np.zeros((0,))


# ----------------- File discovery -----------------
def wav_header(path: str):
    """(format_tag, channels, sample_rate, bits, data_bytes) from the RIFF chunk HEADERS alone: chunk bodies are skipped
    with seek(), so a 30-minute recording costs a few dozen bytes of I/O (torchaudio.info, :133)."""
    with open(path, "rb") as f:
        head = f.read(12)
        if len(head) < 12 or head[:4] != b"RIFF" or head[8:12] != b"WAVE":
            raise ValueError(f"{path}: not a RIFF/WAVE file")
        fmt = data_bytes = None
        end = os.fstat(f.fileno()).st_size
        while True:
            ck = f.read(8)
            if len(ck) < 8:
                break
            cid, size = ck[:4], struct.unpack("<I", ck[4:])[0]
            body_at = f.tell()
            if cid == b"fmt ":
                body = f.read(min(size, 40))
                tag, ch, sr, _br, _ba, bits = struct.unpack("<HHIIHH", body[:16])
                if tag == 0xFFFE and len(body) >= 26:
                    tag = struct.unpack("<H", body[24:26])[0]
                fmt = (tag, ch, sr, bits)
            elif cid == b"data":
                data_bytes = max(0, min(size, end - body_at))      # a truncated file holds what it holds
            f.seek(body_at + size + (size & 1))
    if fmt is None or data_bytes is None:
        raise ValueError(f"{path}: missing fmt/data chunk")
    return (*fmt, data_bytes)


def _wav_num_frames(path: str) -> int:
    """Frames of a WAV file from its header alone (0 when unreadable or when the header describes no samples: zero
    channels / zero bits must count as an empty recording, not raise out of discover_two_files)."""
    try:
        _tag, ch, _sr, bits, nbytes = wav_header(path)
        frame = ch * (bits // 8)
        return nbytes // frame if frame > 0 else 0
    except (OSError, ValueError, ArithmeticError, struct.error):
        return 0


def discover_two_files(root: str, patient_id: str, pattern: str) -> List[str]:
    """The two recordings of a patient (:119-142): files matching `pattern` in any directory below `root` whose path
    contains the patient id; of more than two, the two LONGEST (ties: first in path order), longest first.
    Anything but exactly two is a ValueError."""
    found = sorted(os.path.join(folder, name)
                   for folder, _dirs, names in os.walk(os.path.abspath(root)) if patient_id in folder
                   for name in fnmatch.filter(names, pattern))
    if len(found) > 2:
        frames = {p: _wav_num_frames(p) for p in found}
        found = sorted(found, key=frames.get, reverse=True)[:2]      # stable: equal lengths keep path order
    if len(found) != 2:
        raise ValueError(f"Expected exactly 2 files for patient {patient_id}, found {len(found)}: {found}")
    return found


# ----------------- Aggregation -----------------
def summarize_stage_outputs(stage1_probs: np.ndarray, stage2_probs_or_none: List[Tuple[int, np.ndarray]],
                            stage1_label_order: List[str], stage2_label_order: List[str],
                            stage2_threshold: float = 0.5, use_argmax: bool = False) -> Dict[str, Any]:
    """Per-file summary dict, key for key what the reference writes (:148-195; `use_argmax`: ..._cache.py:243-297).

    Two behaviours of the reference are kept on purpose because utils/aggregate_2stage_results.py consumes these
    numbers: the stage-1 counts come from a PLAIN argmax of the probabilities — the --stage1-threshold gate is not
    re-applied, so "stage1_swallow_windows" can exceed the windows stage 2 evaluated — and the stage-2 ratio divides by
    that argmax count.  Stage-2 rows are taken in window order (a window listed twice counts once, last entry wins)."""
    n = len(stage1_probs)
    is_swallow = np.asarray(stage1_probs).argmax(axis=1) == 1 if n else np.zeros(0, bool)
    swallow = int(is_swallow.sum())
    by_window = {int(i): np.asarray(p) for i, p in stage2_probs_or_none}
    evaluated = [by_window[i] for i in sorted(by_window)]
    if evaluated:
        rows = np.stack(evaluated)
        zenker_mask = rows.argmax(axis=1) == 1 if use_argmax else rows[:, 1] >= stage2_threshold
        healthy_mask = rows.argmax(axis=1) == 0 if use_argmax else rows[:, 1] < stage2_threshold
        zenker, healthy = int(zenker_mask.sum()), int(healthy_mask.sum())
        mean2 = np.mean(evaluated, axis=0).tolist()
    else:
        zenker = healthy = 0
        mean2 = float("nan")            # what np.mean([]) gives the reference when nothing was evaluated
    return {
        "num_windows": int(n),
        "stage1_idle_windows": int(n - swallow),
        "stage1_swallow_windows": swallow,
        "stage1_swallow_ratio": swallow / n if n else 0.0,
        "stage1_mean_probs": np.asarray(stage1_probs).mean(axis=0).tolist() if n else None,
        "stage2_mean_probs_over_swallow": mean2 if swallow else None,
        "stage2_swallow_windows_evaluated": len(evaluated),
        "stage2_healthy_windows": healthy,
        "stage2_zenker_windows": zenker,
        "stage2_zenker_ratio_over_swallow": zenker / swallow if swallow else None,
    }


def classify_recording(audio: np.ndarray, model_s1, fx_s1, model_s2, fx_s2, window_sec: float = 1.0,
                       hop_sec: float = 0.5, stage1_threshold: float = 0.5, stage2_threshold: float = 0.5,
                       stage1_forward_min_prob: Optional[float] = None, stage2_argmax: bool = False,
                       stage1_label_order=("Idle", "Swallow"), stage2_label_order=("Healthy", "Zenker")):
    """The per-file body of main() (:301-348) as ONE library call (zk_two_stage): log-mel once, stage-1 forward,
    on-device gate + compaction, stage-2 forward on the gated windows re-normalised with stage 2's mean/std.
    audio=None: the recording is the context's audio slot (load_audio_to_device) — nothing crosses PCIe but logits.
    Returns (summary dict, s1_probs, s1_preds, stage2_aligned_classes, stage2_results)."""
    ctx = model_s1._ctx
    n_samples = ctx.audio_len() if audio is None else len(audio)
    n, win, hop = window_geometry(n_samples, window_sec, hop_sec)
    model_s1.bind_feature_extractor(fx_s1)
    model_s2.bind_feature_extractor(fx_s2)
    s1_logits, swallow_indices, s2_logits = ctx.two_stage(
        None if audio is None else np.ascontiguousarray(audio, dtype=np.float32), n_samples, 0, hop, win, n,
        np.float32(stage1_threshold), stage1_forward_min_prob)
    return _decide(ctx, s1_logits, swallow_indices, s2_logits, stage1_threshold, stage2_threshold, stage2_argmax,
                   stage1_label_order, stage2_label_order)


def classify_features(store, model_s1, fx_s1, model_s2, fx_s2, stage1_threshold: float = 0.5,
                      stage2_threshold: float = 0.5, stage1_forward_min_prob: Optional[float] = None,
                      stage2_argmax: bool = False, stage1_label_order=("Idle", "Swallow"),
                      stage2_label_order=("Healthy", "Zenker")):
    """classify_recording for a recording whose log-mel comes out of the feature cache (`zkast.cache.CompactFeatures`):
    the store goes into the device slot once, stage 1 runs on it, the gate on the device, stage 2 on the gated windows of
    the SAME slot re-normalised with stage 2's mean / std.  Same return value as classify_recording."""
    ctx = model_s1._ctx
    model_s1.bind_feature_extractor(fx_s1)
    model_s2.bind_feature_extractor(fx_s2)
    n = len(store)
    if n == 0:
        raise RuntimeError("the cached feature store holds no windows")
    store.to_device(ctx)
    s1_logits = model_s1.forward_from_slot(n)
    _probs, swallow_indices = ctx.gate(s1_logits, np.float32(stage1_threshold), stage1_forward_min_prob)
    s2_logits = (model_s2.forward_from_slot(len(swallow_indices), swallow_indices) if len(swallow_indices)
                 else np.zeros((0, 2), np.float32))
    return _decide(ctx, s1_logits, swallow_indices, s2_logits, stage1_threshold, stage2_threshold, stage2_argmax,
                   stage1_label_order, stage2_label_order)


def _decide(ctx, s1_logits, swallow_indices, s2_logits, stage1_threshold, stage2_threshold, stage2_argmax,
            stage1_label_order, stage2_label_order):
    """Logits of the cascade -> (summary dict, s1_probs, s1_preds, stage2_aligned_classes, stage2_results) (:307-348)."""
    s1_probs = ctx.softmax(s1_logits)
    if s1_probs.ndim != 2 or s1_probs.shape[1] != 2:
        raise RuntimeError("Stage1 output shape unexpected; expected (N,2)")
    p_swallow = s1_probs[:, 1]
    s1_preds = s1_probs.argmax(axis=1)
    s1_preds = np.where((s1_preds == 1) & (p_swallow >= stage1_threshold), 1, 0)
    s2_probs = ctx.softmax(s2_logits) if len(swallow_indices) else np.zeros((0, 2), np.float32)
    stage2_results = [(int(g), s2_probs[i]) for i, g in enumerate(swallow_indices)]
    stage2_aligned_classes = np.full(len(s1_preds), -1, dtype=int)
    for gidx, probs in stage2_results:
        if stage2_argmax:
            stage2_aligned_classes[gidx] = int(np.argmax(probs))
        else:
            stage2_aligned_classes[gidx] = 1 if probs[1] >= stage2_threshold else 0
    summary = summarize_stage_outputs(s1_probs, stage2_results, list(stage1_label_order), list(stage2_label_order),
                                      stage2_threshold, stage2_argmax)
    return summary, s1_probs, s1_preds, stage2_aligned_classes, stage2_results


def _extractors_differ_only_in_stats(fx_a, fx_b) -> bool:
    """One compact store serves both stages only when the two extractors differ in nothing but mean / std (each stage
    re-normalises the same un-normalised log-mel: an affine map).  The reference re-uses stage-1 features only for EQUAL
    extractors (..._cache.py:418-422) and otherwise extracts stage 2 on its own; a difference in num_mel_bins, max_length,
    do_normalize, sampling_rate ... must therefore not be papered over."""
    da, db = dict(fx_a.to_dict()), dict(fx_b.to_dict())
    for d in (da, db):
        d.pop("mean", None), d.pop("std", None)
    return da == db


def _length_after_resampling(n_frames: int, sr: int, target_sr: int = SAMPLING_RATE) -> int:
    """samples zk_audio_load leaves in the audio slot: ceil(new * n / orig) with the rates reduced by their gcd"""
    if sr == target_sr:
        return n_frames
    g = int(np.gcd(sr, target_sr))
    orig, new = sr // g, target_sr // g
    return (new * n_frames + orig - 1) // orig


def _cached_features(path: str, ctx, fx, window_sec: float, hop_sec: float, cache_dir: str, refresh: bool, log=print,
                     device: int = 0):
    """Compact log-mel of the recording `path`, through `zkast.cache.FeatureCache`: a store (or a reference `.pt` bundle)
    of this file / window grid / extractor is used when present — the window count then comes from the WAV HEADER and the
    recording is neither decoded nor uploaded; otherwise the file goes to the device once, the device computes the
    log-mel and the entry is written (compact store + reference-format twin)."""
    from . import cache as _cache
    _tag, ch, sr, bits, nbytes = wav_header(path)
    frame = ch * (bits // 8)
    n_samples = _length_after_resampling(nbytes // frame if frame > 0 else 0, sr)
    n, win, hop = window_geometry(n_samples, window_sec, hop_sec)
    fc = _cache.FeatureCache(cache_dir, log=log)
    key = _cache.EntryKey.of(path, window_sec, hop_sec, SAMPLING_RATE, _cache.get_fx_fingerprint(fx))
    store = None if refresh else fc.lookup(key, n, fx)
    if store is None:
        got = load_audio_to_device(path, device=device)
        if got != n_samples:      # (header and decoder disagree: trust the decoder's length for the window grid)
            n, win, hop = window_geometry(got, window_sec, hop_sec)
        ctx.logmel(None, 0, 0, hop, win, n)
        store = _cache.CompactFeatures.from_device(ctx, extractor="zk_logmel")
        fc.store(key, store, fx)
    return store


def aggregate_files(per_file: Dict[str, Dict[str, Any]], files: List[str]) -> Dict[str, Any]:
    """:360-382."""
    vals = list(per_file.values())
    total_windows = int(sum(f["num_windows"] for f in vals))
    total_swallow = sum(f["stage1_swallow_windows"] for f in vals)
    total_zenker = sum(f["stage2_zenker_windows"] for f in vals)
    return {
        "files_used": files,
        "total_windows": total_windows,
        "total_idle_windows": int(sum(f["stage1_idle_windows"] for f in vals)),
        "total_swallow_windows": int(total_swallow),
        "total_swallow_ratio": (total_swallow / max(1, total_windows)),
        "total_swallow_windows_evaluated_stage2": int(sum(f["stage2_swallow_windows_evaluated"] for f in vals)),
        "total_healthy_windows": int(sum(f["stage2_healthy_windows"] for f in vals)),
        "total_zenker_windows": int(total_zenker),
        "overall_zenker_ratio_over_swallow": (total_zenker / total_swallow) if total_swallow else None,
    }


def run_patient(files: List[str], model_s1, fx_s1, model_s2, fx_s2, args_like: Dict[str, Any],
                audios: Optional[List[np.ndarray]] = None) -> Dict[str, Any]:
    """The `output` dict main() writes as <pid>_2stage.json (:384-396), for two files (or in-memory recordings)."""
    per_file = {}
    window_sec, hop_sec = args_like.get("window_sec", 1.0), args_like.get("hop_sec", 0.5)
    decide = (args_like.get("stage1_threshold", 0.5), args_like.get("stage2_threshold", 0.5),
              args_like.get("stage1_forward_min_prob"), args_like.get("stage2_argmax", False))
    cache_dir = None if args_like.get("disable_cache") else args_like.get("feature_cache_dir")
    log = args_like.get("log", print)
    device = getattr(model_s1, "_device", 0)
    if cache_dir and audios is None and not _extractors_differ_only_in_stats(fx_s1, fx_s2):
        log("[cache] the two stages' extractors differ in more than mean / std: one feature store cannot serve both; "
            "running without the feature cache")
        cache_dir = None
    for idx, path in enumerate(files):
        if cache_dir and audios is None:      # the cached variant's options (..._cache.py:361-375): features via the cache
            store = _cached_features(path, model_s1._ctx, fx_s1, window_sec, hop_sec, cache_dir,
                                     bool(args_like.get("refresh_cache")), log, device)
            summary, *_ = classify_features(store, model_s1, fx_s1, model_s2, fx_s2, *decide)
        else:
            if audios is not None:
                audio = audios[idx]
            else:      # file -> device once; decode, resample, log-mel and both forwards read it there
                load_audio_to_device(path, device=device)
                audio = None
            summary, *_ = classify_recording(audio, model_s1, fx_s1, model_s2, fx_s2, window_sec, hop_sec, *decide)
        per_file[f"file_{idx}"] = {"path": path, **summary}
    return {
        "config": {
            "stage1_model_root": args_like.get("stage1_model_root"),
            "stage2_model_root": args_like.get("stage2_model_root"),
            "window_sec": args_like.get("window_sec", 1.0),
            "hop_sec": args_like.get("hop_sec", 0.5),
            "batch_size": args_like.get("batch_size", 128),
            "stage1_threshold": args_like.get("stage1_threshold", 0.5),
            "files": files,
        },
        "per_file": per_file,
        "aggregate": aggregate_files(per_file, files),
    }


# ----------------- CLI (same flags as the reference parser, :201-248, plus the cached variant's two) -------------
def build_arg_parser():
    ap = argparse.ArgumentParser(description="Two-stage AST inference over two long audio files (windowed), MI355X.")
    ap.add_argument("--stage1-model-root")
    ap.add_argument("--stage2-model-root")
    ap.add_argument("--fold", type=int)
    ap.add_argument("--file-a")
    ap.add_argument("--file-b")
    ap.add_argument("--patient-id")
    ap.add_argument("--long-audio-root")
    ap.add_argument("--pattern", default="*.wav")
    ap.add_argument("--window-sec", type=float, default=1.0)
    ap.add_argument("--hop-sec", type=float, default=0.5)
    ap.add_argument("--batch-size", type=int, default=128)
    ap.add_argument("--stage1-threshold", type=float, default=0.5)
    ap.add_argument("--stage2-threshold", type=float, default=0.5)
    ap.add_argument("--stage1-forward-min-prob", type=float, default=None)
    ap.add_argument("--stage2-argmax", action="store_true")
    ap.add_argument("--output-json")
    ap.add_argument("--show-first-n", type=int, default=5)
    ap.add_argument("--compute-mode", default=DEFAULT_COMPUTE_MODE, choices=["f16", "f16c8", "f16x3", "f16mix"])
    # the cached variant's cache options (..._cache.py:361-375); here the cache is OFF unless a directory is given — the
    # device log-mel costs 0.5 ms per 1 024 windows, the cache only matters for exchanging features with the reference
    ap.add_argument("--feature-cache-dir", default=None,
                    help="Directory of cached features (this build's compact stores and the reference's .pt bundles).")
    ap.add_argument("--disable-cache", action="store_true", help="Ignore --feature-cache-dir.")
    ap.add_argument("--refresh-cache", action="store_true", help="Recompute and overwrite existing cache entries.")
    return ap


def _files_of(args) -> List[str]:
    """The two recordings of the run: given explicitly, or discovered below --long-audio-root for --patient-id."""
    if args.file_a and args.file_b:
        return [args.file_a, args.file_b]
    if args.patient_id and args.long_audio_root:
        return discover_two_files(args.long_audio_root, args.patient_id, args.pattern)
    raise ValueError("Provide either --file-a & --file-b or (--patient-id and --long-audio-root).")


def _model_roots_of(args) -> Tuple[str, str]:
    """(stage-1 root, stage-2 root): explicit flags win; --fold k fills a missing one with the training scripts' layout
    `runs/ast_classifier_stage<s>/fold<k>/best` below the working directory."""
    roots = []
    for stage, given in ((1, args.stage1_model_root), (2, args.stage2_model_root)):
        if not given and args.fold is not None:
            given = os.path.join(os.getcwd(), "runs", f"ast_classifier_stage{stage}", f"fold{args.fold}", "best")
        if not given:
            raise ValueError("Model roots must be provided either explicitly or via --fold.")
        roots.append(given)
    return roots[0], roots[1]


def main(argv=None):
    """The reference script's command line (same flags, same `<pid>_2stage.json`) on the HIP path."""
    args = build_arg_parser().parse_args(argv)
    if min(args.window_sec, args.hop_sec) <= 0:
        raise ValueError("window-sec and hop-sec must be > 0")
    files = _files_of(args)
    args.stage1_model_root, args.stage2_model_root = _model_roots_of(args)
    print("Recordings:", *(f"\n  {tag}: {path}" for tag, path in zip("AB", files)))
    if args.hop_sec > args.window_sec:
        print("[WARN] hop-sec exceeds window-sec: consecutive windows leave gaps.")
    stages = [load_stage_model(root, labels, slot, args.compute_mode)
              for slot, (root, labels) in enumerate(((args.stage1_model_root, ["Idle", "Swallow"]),
                                                     (args.stage2_model_root, ["Healthy", "Zenker"])))]
    (fx_s1, model_s1), (fx_s2, model_s2) = stages
    output = run_patient(files, model_s1, fx_s1, model_s2, fx_s2, vars(args))
    for name, rec in output["per_file"].items():
        print(f"{name}: {rec['num_windows']} windows, swallow {rec['stage1_swallow_windows']}, zenker {rec['stage2_zenker_windows']}")
    target = args.output_json or (os.path.join("outputs", f"{args.patient_id}_2stage.json") if args.patient_id else None)
    if target:
        os.makedirs(os.path.dirname(target) or ".", exist_ok=True)
        with open(target, "w") as f:
            json.dump(output, f, indent=2)
        print(f"Saved JSON: {target}")
    return output


if __name__ == "__main__":
    main()

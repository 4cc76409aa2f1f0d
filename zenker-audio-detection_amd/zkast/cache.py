"""Feature cache of the cached variant (src/test_long_audio_windows_2stage_cache.py:84-208): same fingerprint, cache
key, `.pt` bundle layout ({"metadata": {...}, "features": (N,1024,128) float32}) and metadata check, so caches written
by the reference and by this build are interchangeable.  On MI355X the log-mel costs ~0.1 ms per window, so the cache
only matters for interoperability; `classify_recording` never needs it."""
from __future__ import annotations

import hashlib
import json
import os
from typing import Any, Dict, List, Optional

import numpy as np

from . import lib as _lib
from .pipeline import SAMPLING_RATE, batch_iter


def get_fx_fingerprint(fx) -> str:
    """:84-86."""
    return hashlib.sha256(json.dumps(fx.to_dict(), sort_keys=True).encode("utf-8")).hexdigest()


def build_cache_path(cache_dir: str, audio_path: str, window_sec: float, hop_sec: float, sr: int,
                     fx_fingerprint: str) -> str:
    """:89-103."""
    audio_abs = os.path.abspath(audio_path)
    audio_stats = f"{os.path.getsize(audio_abs)}_{int(os.path.getmtime(audio_abs))}"
    key = f"{audio_abs}|{window_sec}|{hop_sec}|{sr}|{fx_fingerprint}|{audio_stats}"
    digest = hashlib.sha256(key.encode("utf-8")).hexdigest()[:16]
    base = os.path.splitext(os.path.basename(audio_abs))[0]
    return os.path.join(cache_dir, f"{base}_{digest}.pt")


def build_base_metadata(audio_path: str, window_sec: float, hop_sec: float, num_windows: int, sr: int,
                        fx_fingerprint: str) -> Dict[str, Any]:
    """:106-124."""
    audio_abs = os.path.abspath(audio_path)
    return {
        "audio_path": audio_abs,
        "audio_size": os.path.getsize(audio_abs),
        "audio_mtime": int(os.path.getmtime(audio_abs)),
        "window_sec": window_sec,
        "hop_sec": hop_sec,
        "num_windows": num_windows,
        "sampling_rate": sr,
        "extractor_fingerprint": fx_fingerprint,
    }


def compute_features(fx, windows: List[np.ndarray], batch_size: int):
    """:127-139 -> torch.FloatTensor (N,1024,128) on the CPU."""
    import torch
    name = fx.model_input_names[0]
    chunks = [fx(batch, sampling_rate=SAMPLING_RATE, return_tensors="pt")[name] for batch in batch_iter(windows, batch_size)]
    if not chunks:
        raise RuntimeError("Feature extraction yielded no data; check window setup.")
    return torch.cat(chunks, dim=0).to(torch.float32).contiguous()


def load_or_compute_features(audio_path: str, windows: List[np.ndarray], fx, window_sec: float, hop_sec: float,
                             batch_size: int, cache_dir: Optional[str], disable_cache: bool = False,
                             refresh_cache: bool = False, stage_label: str = "stage1", log=print):
    """:142-192."""
    import torch
    fp = get_fx_fingerprint(fx)
    base_meta = build_base_metadata(audio_path, window_sec, hop_sec, len(windows), SAMPLING_RATE, fp)
    if disable_cache or not cache_dir:
        return compute_features(fx, windows, batch_size)
    os.makedirs(cache_dir, exist_ok=True)
    cache_path = build_cache_path(cache_dir, audio_path, window_sec, hop_sec, SAMPLING_RATE, fp)
    if not refresh_cache and os.path.exists(cache_path):
        try:
            bundle = torch.load(cache_path, map_location="cpu")
            metadata = bundle.get("metadata", {})
            if all(metadata.get(k) == v for k, v in base_meta.items()):
                log(f"[cache:{stage_label}] Loaded {cache_path}")
                return bundle["features"].to(torch.float32).contiguous()
            log(f"[cache:{stage_label}] Metadata mismatch for {cache_path}; recomputing.")
        except Exception as exc:
            log(f"[cache:{stage_label}] Failed to load {cache_path}: {exc}; recomputing.")
    features = compute_features(fx, windows, batch_size)
    full_meta = dict(base_meta)
    full_meta["feature_shape"] = list(features.shape)
    try:
        torch.save({"metadata": full_meta, "features": features.cpu()}, cache_path)
        log(f"[cache:{stage_label}] Saved {cache_path}")
    except Exception as exc:
        log(f"[cache:{stage_label}] Failed to save {cache_path}: {exc}")
    return features


def forward_probs_from_features(model, features, batch_size: int) -> np.ndarray:
    """:198-208: features (N,1024,128) tensor/array -> (N,2) float32 softmax probabilities; empty -> zeros((0,0))."""
    ctx = _lib.get_context(getattr(model, "_device", 0))
    n = int(features.shape[0])
    probs_all = []
    for start in range(0, n, batch_size):
        logits = model(features[start:start + batch_size]).logits
        probs_all.append(ctx.softmax(np.asarray(logits)))
    return np.concatenate(probs_all, axis=0) if probs_all else np.zeros((0, 0))

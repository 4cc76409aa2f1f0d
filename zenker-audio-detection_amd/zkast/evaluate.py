"""Snippet-level evaluation loops on the accelerated classes: the same
``feature_extractor(wavs, sampling_rate=16000, return_tensors="pt") -> model(feats).logits -> softmax`` contract the
reference uses for ROC/PR analysis (utils/analyze_ROC_PR_stage1.py:116-191) and for the cross-validation test runs
(src/test_trained_model_stage1_cv.py:101-169, where a transformers ``Trainer.predict`` with
``per_device_eval_batch_size=8`` yields ``predictions`` = logits and ``label_ids``), at small batch and with snippets
of unequal length.  Plotting, bootstrap CIs and W&B stay in the reference; what is replaced is the part that runs the
network.  Host code only; the arithmetic is the HIP library's (no CPU fallback)."""
from __future__ import annotations

import os
from typing import List, Sequence, Tuple

import numpy as np

from . import lib as _lib
from .lib import DEFAULT_COMPUTE_MODE
from .feature_extraction import ZkASTFeatureExtractor
from .modeling import ZkASTConfig, ZkASTForAudioClassification
from .pipeline import SAMPLING_RATE, load_audio


def load_split(data_dir: str, fold: int, preferred_split: str) -> Tuple[List, List, str]:
    """(snippets, labels, split actually used) of one fold.  Files are `<split>_x_fold<k>.npy` (object array of audio
    payloads) and `<split>_y_fold<k>.npy`; asking for "val" falls back to "test" when the fold has no validation files,
    anything else reads "test" (utils/analyze_ROC_PR_stage1.py:116-129)."""
    for split in ((preferred_split, "test") if preferred_split == "val" else ("test",)):
        xs, ys = (os.path.join(data_dir, f"{split}_{axis}_fold{fold}.npy") for axis in "xy")
        if os.path.isfile(xs) and os.path.isfile(ys):
            return np.load(xs, allow_pickle=True).tolist(), np.load(ys).astype(int).tolist(), split
    raise FileNotFoundError(f"No {preferred_split} or test split found for fold {fold} in {data_dir}.")


_DICT_AUDIO_KEYS = ("array", "audio", "values")
_DICT_RATE_KEYS = ("sampling_rate", "sampling_rate_hz")


def to_waveform(entry, device: int = 0) -> np.ndarray:
    """One dataset payload -> mono float32 at 16 kHz (:132-155).  Payload kinds: an ndarray (taken as 16 kHz), a dict with
    the samples under "array" | "audio" | "values" and optionally their rate under "sampling_rate" | "sampling_rate_hz"
    (resampled on the GPU when it differs), or a path to a WAV file (decoded and resampled on the GPU)."""
    if isinstance(entry, str):
        return load_audio(entry, SAMPLING_RATE, device)
    if isinstance(entry, np.ndarray):
        return entry.astype(np.float32)
    if not isinstance(entry, dict):
        raise TypeError(f"Unsupported audio payload type: {type(entry)}")
    samples = next((entry[k] for k in _DICT_AUDIO_KEYS if entry.get(k) is not None), None)
    if samples is None:
        raise ValueError("Unsupported dict payload for audio sample.")
    wav = np.ascontiguousarray(samples, dtype=np.float32)
    rate = int(next((entry[k] for k in _DICT_RATE_KEYS if entry.get(k)), SAMPLING_RATE))
    return wav if rate == SAMPLING_RATE else _lib.get_context(device).resample(wav, rate, SAMPLING_RATE)


def batched(iterable: Sequence, batch_size: int):
    """Consecutive slices of at most `batch_size` items (:158-160)."""
    return (iterable[lo:lo + batch_size] for lo in range(0, len(iterable), batch_size))


def predict_logits(model: ZkASTForAudioClassification, feature_extractor: ZkASTFeatureExtractor, X: Sequence,
                   batch_size: int = 8, device: int = 0) -> np.ndarray:
    """The network part of ``Trainer.predict`` (test_trained_model_stage1_cv.py:101-160): (N, num_labels) float32
    logits in dataset order, `batch_size` snippets per forward."""
    name = feature_extractor.model_input_names[0]
    out = []
    for batch_entries in batched(X, batch_size):
        wavs = [to_waveform(e, device) for e in batch_entries]
        inputs = feature_extractor(wavs, sampling_rate=SAMPLING_RATE, return_tensors="np", padding=True)
        out.append(np.asarray(model(inputs[name]).logits, dtype=np.float32))
    n_labels = getattr(model, "num_labels", 2)
    return np.concatenate(out) if out else np.zeros((0, n_labels), dtype=np.float32)


def softmax(logits: np.ndarray) -> np.ndarray:
    z = logits - logits.max(axis=1, keepdims=True)
    e = np.exp(z)
    return (e / e.sum(axis=1, keepdims=True)).astype(np.float32)


def run_inference(model_dir: str, X: Sequence, batch_size: int, stage: int = 0, compute_mode=DEFAULT_COMPUTE_MODE,
                  device: int = 0) -> np.ndarray:
    """:163-191 — scores = softmax(logits)[:, 1] (probability of class 1) for every snippet of X."""
    feature_extractor = ZkASTFeatureExtractor.from_pretrained(model_dir, device=device)
    config = ZkASTConfig.from_pretrained(model_dir)
    model = ZkASTForAudioClassification.from_pretrained(model_dir, config=config, stage=stage,
                                                        compute_mode=compute_mode, device=device)
    model.eval()
    logits = predict_logits(model, feature_extractor, X, batch_size, device)
    return softmax(logits)[:, 1] if len(logits) else np.zeros((0,), dtype=np.float32)


def evaluate_predictions(logits: np.ndarray, y_true: Sequence[int], n_classes: int):
    """argmax predictions + confusion matrix (rows = true class, columns = predicted; what sklearn's
    ``confusion_matrix(y_true, y_pred, labels=range(n))`` returns at test_trained_model_stage1_cv.py:162-163)."""
    y_pred = np.asarray(logits).argmax(axis=1) if len(logits) else np.zeros((0,), dtype=np.int64)
    cm = np.zeros((n_classes, n_classes), dtype=np.int64)
    for t, p in zip(y_true, y_pred):
        cm[int(t), int(p)] += 1
    return y_pred, cm

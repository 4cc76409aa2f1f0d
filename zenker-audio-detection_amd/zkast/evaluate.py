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
from .feature_extraction import ZkASTFeatureExtractor
from .modeling import ZkASTConfig, ZkASTForAudioClassification
from .pipeline import SAMPLING_RATE, load_audio


def load_split(data_dir: str, fold: int, preferred_split: str) -> Tuple[List, List, str]:
    """utils/analyze_ROC_PR_stage1.py:116-129 — `<split>_x_fold{k}.npy` / `<split>_y_fold{k}.npy`, val falls back to test."""
    candidates = [preferred_split, "test"] if preferred_split == "val" else ["test"]
    for split in candidates:
        x_path = os.path.join(data_dir, f"{split}_x_fold{fold}.npy")
        y_path = os.path.join(data_dir, f"{split}_y_fold{fold}.npy")
        if os.path.exists(x_path) and os.path.exists(y_path):
            X = np.load(x_path, allow_pickle=True).tolist()
            y = np.load(y_path).astype(int).tolist()
            return X, y, split
    raise FileNotFoundError(f"No {preferred_split} or test split found for fold {fold} in {data_dir}.")


def to_waveform(entry, device: int = 0) -> np.ndarray:
    """:132-155 — ndarray | {"array"|"audio"|"values", "sampling_rate"|"sampling_rate_hz"} | path -> mono float32 @ 16 kHz
    (resampling runs on the GPU, files go through the package's own RIFF reader)."""
    if isinstance(entry, np.ndarray):
        return entry.astype(np.float32)
    if isinstance(entry, dict):
        arr = next((entry[k] for k in ("array", "audio", "values") if entry.get(k) is not None), None)
        if arr is None:
            raise ValueError("Unsupported dict payload for audio sample.")
        arr = np.asarray(arr, dtype=np.float32)
        sr = entry.get("sampling_rate") or entry.get("sampling_rate_hz") or SAMPLING_RATE
        if sr != SAMPLING_RATE:
            arr = _lib.get_context(device).resample(np.ascontiguousarray(arr), int(sr), SAMPLING_RATE)
        return arr
    if isinstance(entry, str):
        return load_audio(entry, SAMPLING_RATE, device)
    raise TypeError(f"Unsupported audio payload type: {type(entry)}")


def batched(iterable: Sequence, batch_size: int):
    """:158-160"""
    for i in range(0, len(iterable), batch_size):
        yield iterable[i : i + batch_size]


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


def run_inference(model_dir: str, X: Sequence, batch_size: int, stage: int = 0, compute_mode="f16c8",
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

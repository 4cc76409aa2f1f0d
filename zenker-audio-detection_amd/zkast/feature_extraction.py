"""Drop-in for ``transformers.ASTFeatureExtractor`` on the reference's hot path
(``fx(batch, sampling_rate=16000, return_tensors="pt")`` at src/test_long_audio_windows_2stage.py:108).

Same constructor arguments, ``model_input_names``, ``to_dict()`` keys, sampling-rate ``ValueError`` and output
contract (``input_values``: (B, 1024, 128) float32, real rows first, zero rows normalised to ``-mean/(2 std)``) as
``$TF/models/audio_spectrogram_transformer/feature_extraction_audio_spectrogram_transformer.py:68-234``; the arithmetic
runs in ``logmel.hip`` on the GPU (float64, as the numpy branch of the original).
"""
from __future__ import annotations

import json
import os

import numpy as np

from . import lib as _lib


class BatchFeature(dict):
    """Minimal stand-in for transformers.BatchFeature: mapping with attribute access and ``.to()``."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:  # pragma: no cover
            raise AttributeError(k) from e

    def to(self, device):
        for k, v in list(self.items()):
            if hasattr(v, "to"):
                self[k] = v.to(device)
        return self


class ZkASTFeatureExtractor:
    model_input_names = ["input_values", "attention_mask"]

    def __init__(self, feature_size=1, sampling_rate=16000, num_mel_bins=128, max_length=1024, padding_value=0.0,
                 do_normalize=True, mean=-4.2677393, std=4.5689974, return_attention_mask=False, padding_side="right",
                 device=0, **kwargs):
        if num_mel_bins != 128 or max_length != 1024 or sampling_rate != 16000:
            raise ValueError("ZkASTFeatureExtractor is specialised to sampling_rate=16000, num_mel_bins=128, "
                             "max_length=1024 (the AST configuration the reference trains and runs)")
        self.feature_size = feature_size
        self.sampling_rate = sampling_rate
        self.num_mel_bins = num_mel_bins
        self.max_length = max_length
        self.padding_value = padding_value
        self.do_normalize = do_normalize
        self.mean = mean
        self.std = std
        self.return_attention_mask = return_attention_mask
        self.padding_side = padding_side
        self._device = device
        self._extra = {k: v for k, v in kwargs.items() if k not in ("feature_extractor_type", "processor_class")}

    # ---- persistence (preprocessor_config.json written by feature_extractor.save_pretrained,
    #      src/train_ast_stage1_cross_validation.py:524) ----
    @classmethod
    def from_pretrained(cls, model_root: str, **kwargs):
        path = os.path.join(model_root, "preprocessor_config.json")
        if not os.path.isfile(path):
            raise OSError(f"{path} not found (ZkASTFeatureExtractor.from_pretrained only reads local directories)")
        with open(path) as f:
            cfg = json.load(f)
        cfg.update(kwargs)
        return cls(**cfg)

    def to_dict(self) -> dict:
        d = {
            "do_normalize": self.do_normalize,
            "feature_extractor_type": "ASTFeatureExtractor",
            "feature_size": self.feature_size,
            "max_length": self.max_length,
            "mean": self.mean,
            "num_mel_bins": self.num_mel_bins,
            "padding_side": self.padding_side,
            "padding_value": self.padding_value,
            "return_attention_mask": self.return_attention_mask,
            "sampling_rate": self.sampling_rate,
            "std": self.std,
        }
        return d

    def save_pretrained(self, model_root: str):
        os.makedirs(model_root, exist_ok=True)
        with open(os.path.join(model_root, "preprocessor_config.json"), "w") as f:
            json.dump(self.to_dict(), f, indent=2, sort_keys=True)

    def __repr__(self):
        return f"ZkASTFeatureExtractor {json.dumps(self.to_dict(), indent=2, sort_keys=True)}"

    # ---- the call ----
    def _check_rate(self, sampling_rate):
        if sampling_rate is not None and sampling_rate != self.sampling_rate:
            raise ValueError(
                f"The model corresponding to this feature extractor: {self} was trained using a sampling rate of"
                f" {self.sampling_rate}. Please make sure that the provided `raw_speech` input was sampled with"
                f" {self.sampling_rate} and not {sampling_rate}."
            )

    @staticmethod
    def _as_batch(raw_speech):
        is_batched_numpy = isinstance(raw_speech, np.ndarray) and raw_speech.ndim > 1
        if is_batched_numpy and raw_speech.ndim > 2:
            raise ValueError("Only mono-channel audio is supported for input to ZkASTFeatureExtractor")
        is_batched = is_batched_numpy or (
            isinstance(raw_speech, (list, tuple)) and len(raw_speech) > 0
            and isinstance(raw_speech[0], (np.ndarray, tuple, list))
        )
        if is_batched:
            return [np.asarray(s, dtype=np.float32) for s in raw_speech]
        return [np.asarray(raw_speech, dtype=np.float32)]

    def __call__(self, raw_speech, sampling_rate=None, return_tensors=None, **kwargs):
        self._check_rate(sampling_rate)
        speech = [np.squeeze(s) if s.ndim > 1 else s for s in self._as_batch(raw_speech)]
        ctx = _lib.get_context(self._device)
        out = np.empty((len(speech), self.max_length, self.num_mel_bins), dtype=np.float32)
        # group equal-length waveforms: each group is one kernel launch over a (B, L) block
        by_len = {}
        for i, s in enumerate(speech):
            by_len.setdefault(int(s.shape[0]), []).append(i)
        for length, ids in by_len.items():
            if length < 400:
                raise ValueError(f"waveform of {length} samples is shorter than one 25 ms frame (400 samples)")
            block = np.ascontiguousarray(np.stack([speech[i] for i in ids]).reshape(-1))
            ctx.logmel(block, block.size, 0, length, length, len(ids))
            part = np.empty((len(ids), self.max_length, self.num_mel_bins), dtype=np.float32)
            ctx.features_expand(self.mean, self.std, self.do_normalize, part)
            out[ids] = part
        if return_tensors is None:
            return BatchFeature({"input_values": [o for o in out]})
        if return_tensors == "np":
            return BatchFeature({"input_values": out})
        if return_tensors == "pt":
            import torch
            return BatchFeature({"input_values": torch.from_numpy(out)})
        raise ValueError(f"unsupported return_tensors={return_tensors!r}")

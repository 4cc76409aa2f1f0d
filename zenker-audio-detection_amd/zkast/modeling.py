"""Drop-in for ``transformers.ASTForAudioClassification`` / ``ASTConfig`` on the reference's hot path
(``model(feats).logits`` at src/test_long_audio_windows_2stage.py:110; ``from_pretrained(<local dir>)`` at :89-98).

The forward pass is the hand-written HIP path behind ``zk_ast_forward``; this file only reads ``config.json`` and the
checkpoint (safetensors or torch ``.bin``; both the 4.x and 5.x key schemes) into host buffers and hands them over.
PyTorch is used for nothing but that file reading and for carrying tensors in and out.
"""
from __future__ import annotations

import json
import os

import numpy as np

from . import lib as _lib
from .lib import DEFAULT_COMPUTE_MODE

_CFG_KEYS = ("hidden_size", "num_hidden_layers", "num_attention_heads", "intermediate_size", "patch_size",
             "frequency_stride", "time_stride", "max_length", "num_mel_bins", "layer_norm_eps")


class ZkASTConfig:
    """The fields of ASTConfig the path uses ($TF/.../configuration_audio_spectrogram_transformer.py:50-64)."""

    def __init__(self, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                 hidden_act="gelu", layer_norm_eps=1e-12, patch_size=16, qkv_bias=True, frequency_stride=10,
                 time_stride=10, max_length=1024, num_mel_bins=128, num_labels=2, id2label=None, label2id=None,
                 **kwargs):
        self.hidden_size = hidden_size
        self.num_hidden_layers = num_hidden_layers
        self.num_attention_heads = num_attention_heads
        self.intermediate_size = intermediate_size
        self.hidden_act = hidden_act
        self.layer_norm_eps = layer_norm_eps
        self.patch_size = patch_size
        self.qkv_bias = qkv_bias
        self.frequency_stride = frequency_stride
        self.time_stride = time_stride
        self.max_length = max_length
        self.num_mel_bins = num_mel_bins
        if id2label is not None:
            num_labels = len(id2label)
        self.num_labels = num_labels
        self.id2label = id2label or {i: f"LABEL_{i}" for i in range(num_labels)}
        self.label2id = label2id or {v: k for k, v in self.id2label.items()}
        if hidden_act != "gelu" or not qkv_bias:
            raise ValueError("zkast implements the AST configuration of the reference: hidden_act='gelu', qkv_bias=True")

    @classmethod
    def from_pretrained(cls, model_root: str, **kwargs):
        path = os.path.join(model_root, "config.json")
        if not os.path.isfile(path):
            raise OSError(f"{path} not found (ZkASTConfig.from_pretrained only reads local directories)")
        with open(path) as f:
            d = json.load(f)
        d.update(kwargs)
        if "id2label" in d and d["id2label"] is not None:
            d["id2label"] = {int(k): v for k, v in d["id2label"].items()}
        return cls(**d)

    def to_dict(self):
        d = {k: getattr(self, k) for k in _CFG_KEYS}
        d.update(num_labels=self.num_labels, id2label=self.id2label, label2id=self.label2id, hidden_act="gelu",
                 qkv_bias=True, model_type="audio-spectrogram-transformer")
        return d


class SequenceClassifierOutput:
    def __init__(self, logits):
        self.logits = logits
        self.loss = None

    def __getitem__(self, i):
        return (self.logits,)[i]


def _read_checkpoint(model_root: str) -> dict:
    st = os.path.join(model_root, "model.safetensors")
    if os.path.isfile(st):
        from safetensors.numpy import load_file
        try:
            return load_file(st)
        except Exception:  # bf16 checkpoints are not representable in numpy
            from safetensors.torch import load_file as load_pt
            return load_pt(st)
    pt = os.path.join(model_root, "pytorch_model.bin")
    if os.path.isfile(pt):
        import torch
        return torch.load(pt, map_location="cpu", weights_only=True)
    raise OSError(f"no model.safetensors or pytorch_model.bin in {model_root}")


class ZkASTForAudioClassification:
    """One stage of the cascade, resident on one GPU.  ``stage`` selects the library's weight slot (0: Idle/Swallow,
    1: Healthy/Zenker)."""

    main_input_name = "input_values"

    def __init__(self, config: ZkASTConfig, state_dict: dict, stage: int = 0, compute_mode=DEFAULT_COMPUTE_MODE, device: int = 0,
                 fx_mean: float = -4.2677393, fx_std: float = 4.5689974):
        self.config = config
        self.stage = int(stage)
        self.compute_mode = compute_mode
        self._device = device
        self._ctx = _lib.get_context(device)
        self._ctx.load_model(self.stage, state_dict, config.to_dict(), fx_mean, fx_std, compute_mode)
        self.num_labels = config.num_labels

    @classmethod
    def from_pretrained(cls, model_root: str, config: ZkASTConfig | None = None, stage: int = 0,
                        compute_mode=DEFAULT_COMPUTE_MODE, device: int = 0, **kwargs):
        if config is None:
            config = ZkASTConfig.from_pretrained(model_root)
        sd = _read_checkpoint(model_root)
        mean, std = -4.2677393, 4.5689974
        pp = os.path.join(model_root, "preprocessor_config.json")
        if os.path.isfile(pp):
            with open(pp) as f:
                p = json.load(f)
            mean, std = p.get("mean", mean), p.get("std", std)
        return cls(config, sd, stage=stage, compute_mode=compute_mode, device=device, fx_mean=mean, fx_std=std)

    # torch.nn.Module look-alikes used by the reference script
    def to(self, *_a, **_k):
        return self

    def eval(self):
        return self

    def bind_feature_extractor(self, fx):
        """Tell the library which mean/std normalise the feature slot for this stage (fused path)."""
        self._ctx.set_fx(self.stage, fx.mean, fx.std)
        return self

    def set_compute_mode(self, mode):
        self.compute_mode = mode
        self._ctx.set_compute_mode(self.stage, mode)

    def set_layer_modes(self, modes):
        """per-layer compute modes (ZK_F16MIX): e.g. ["f16x3"] * 4 + ["f16c8"] * 8"""
        self.compute_mode = "f16mix"
        self._ctx.set_layer_modes(self.stage, list(modes))

    def __call__(self, input_values=None, **kwargs):
        return self.forward(input_values, **kwargs)

    def forward(self, input_values=None, **kwargs):
        if input_values is None:
            raise ValueError("input_values is required")
        is_torch = type(input_values).__module__.startswith("torch")
        if is_torch:
            import torch
            x = input_values.detach()
            if x.dtype != torch.float32:
                x = x.float()
            x = x.contiguous()
            shape = tuple(x.shape)
        else:
            x = np.ascontiguousarray(input_values, dtype=np.float32)
            shape = x.shape
        if len(shape) != 3 or shape[1] != 1024 or shape[2] != 128:
            raise ValueError(f"input_values must be (batch, 1024, 128), got {shape}")
        B = shape[0]
        if is_torch:
            import torch
            logits = torch.empty((B, self.num_labels), dtype=torch.float32, device=x.device)
        else:
            logits = np.empty((B, self.num_labels), dtype=np.float32)
        self._ctx.ast_forward(self.stage, x, None, B, logits)
        return SequenceClassifierOutput(logits)

    def forward_from_slot(self, n_windows: int, win_idx=None) -> np.ndarray:
        """logits for windows of the library's feature slot (no (B,1024,128) tensor is ever materialised)."""
        B = int(n_windows if win_idx is None else len(win_idx))
        logits = np.empty((B, self.num_labels), dtype=np.float32)
        idx = None if win_idx is None else np.ascontiguousarray(win_idx, dtype=np.int32)
        self._ctx.ast_forward(self.stage, None, idx, B, logits)
        return logits

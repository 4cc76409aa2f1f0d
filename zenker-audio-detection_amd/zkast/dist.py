"""Multi-GPU sharding of the cascade: one process per GPU, windows are the independent unit.

The reference is single-process / single-device; a long recording is a batch of independent 1 s windows
(src/test_long_audio_windows_2stage.py:62-75), so the N windows are partitioned into contiguous ranges, each rank
runs stage 1 on its range, the per-window logits (N x 2 fp32 — a few KB) are all-gathered over RCCL/xGMI
(`torch.distributed`, backend "nccl" on the GPU box, "gloo" in the CPU tests), every rank derives the identical
gate, the K gated windows are re-partitioned evenly (swallows cluster in time, so re-using the stage-1 partition would
imbalance stage 2) and the stage-2 logits are all-gathered the same way.  No other data-path collective exists.

The compute is injected as two callables so that the partition / gather / gate logic is testable on CPU (gloo,
world_size 2) without a GPU; `ZkShardedCascade` binds them to the HIP path.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import numpy as np


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """contiguous [lo, hi) of rank; ranges differ by at most one element and cover [0, n)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _dist():
    import torch.distributed as dist
    return dist


def all_gather_rows(local: np.ndarray, n_total: int, world: int, device=None) -> np.ndarray:
    """all-gather of contiguous row shards produced with shard_range; returns (n_total, cols) on every rank.
    Shards are padded to the largest count so that ONE fixed-size collective is issued."""
    import torch
    dist = _dist()
    cols = local.shape[1] if local.ndim == 2 else 2
    if world == 1 or not (dist.is_available() and dist.is_initialized()):
        return np.ascontiguousarray(local, dtype=np.float32).reshape(n_total, cols)
    per = (n_total + world - 1) // world
    buf = torch.zeros((per, cols), dtype=torch.float32)
    if local.shape[0]:
        buf[: local.shape[0]] = torch.from_numpy(np.ascontiguousarray(local, dtype=np.float32))
    if device is not None:
        buf = buf.to(device)
    out = torch.empty((world * per, cols), dtype=torch.float32, device=buf.device)
    dist.all_gather_into_tensor(out, buf)
    out = out.cpu().numpy().reshape(world, per, cols)
    parts = []
    for r in range(world):
        lo, hi = shard_range(n_total, r, world)
        parts.append(out[r, : hi - lo])
    return np.concatenate(parts, axis=0) if parts else np.zeros((0, cols), np.float32)


def softmax_np(logits: np.ndarray) -> np.ndarray:
    z = logits - logits.max(axis=1, keepdims=True)
    e = np.exp(z)
    return (e / e.sum(axis=1, keepdims=True)).astype(np.float32)


def gate_indices(s1_probs: np.ndarray, thr1: float, fwd_min_prob: Optional[float] = None) -> np.ndarray:
    """src/test_long_audio_windows_2stage.py:312-320 (+ ..._cache.py:471-478)."""
    p = s1_probs[:, 1]
    pred = s1_probs.argmax(axis=1)
    pred = np.where((pred == 1) & (p >= np.float32(thr1)), 1, 0)
    idx = np.where(pred == 1)[0]
    if fwd_min_prob is not None:
        idx = idx[p[idx] >= np.float32(fwd_min_prob)]
    return idx.astype(np.int32)


def sharded_cascade(n_windows: int, stage_logits: Callable[[int, np.ndarray], np.ndarray], rank: int, world: int,
                    thr1: float, fwd_min_prob: Optional[float] = None, device=None,
                    softmax: Callable[[np.ndarray], np.ndarray] = softmax_np):
    """stage_logits(stage, win_idx int32[]) -> (len(win_idx), 2) logits of those windows, computed locally.
    Returns (s1_logits (N,2), swallow_idx (K,), s2_logits (K,2)) — identical on every rank."""
    lo, hi = shard_range(n_windows, rank, world)
    mine = np.arange(lo, hi, dtype=np.int32)
    l1 = stage_logits(0, mine) if hi > lo else np.zeros((0, 2), np.float32)
    s1 = all_gather_rows(l1, n_windows, world, device)
    idx = gate_indices(softmax(s1) if n_windows else np.zeros((0, 2), np.float32), thr1, fwd_min_prob)
    k = int(idx.shape[0])
    klo, khi = shard_range(k, rank, world)
    l2 = stage_logits(1, idx[klo:khi]) if khi > klo else np.zeros((0, 2), np.float32)
    s2 = all_gather_rows(l2, k, world, device) if k else np.zeros((0, 2), np.float32)
    return s1, idx, s2


class ZkShardedCascade:
    """Binds sharded_cascade to the HIP path for one recording: every rank computes the (cheap, 7.6 MFLOP/window)
    log-mel of the whole recording into its own feature slot, then runs the two AST forwards on its shards."""

    def __init__(self, model_s1, fx_s1, model_s2, fx_s2, rank: int, world: int, device=None):
        self.m = (model_s1, model_s2)
        self.fx = (fx_s1, fx_s2)
        self.rank, self.world, self.device = rank, world, device

    def __call__(self, audio: np.ndarray, window_sec=1.0, hop_sec=0.5, thr1=0.5, fwd_min_prob=None):
        from .pipeline import window_geometry
        n, win, hop = window_geometry(len(audio), window_sec, hop_sec)
        ctx = self.m[0]._ctx
        for m, fx in zip(self.m, self.fx):
            m.bind_feature_extractor(fx)
        ctx.logmel(np.ascontiguousarray(audio, dtype=np.float32), len(audio), 0, hop, win, n)

        def stage_logits(stage, win_idx):
            return self.m[stage].forward_from_slot(len(win_idx), win_idx)

        return sharded_cascade(n, stage_logits, self.rank, self.world, thr1, fwd_min_prob, self.device, ctx.softmax)

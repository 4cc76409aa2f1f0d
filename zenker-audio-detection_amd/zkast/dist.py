"""Multi-GPU sharding of the cascade: one process per GPU, windows are the independent unit.

The reference is single-process / single-device; a long recording is a batch of independent 1 s windows
(src/test_long_audio_windows_2stage.py:62-75), so the N windows are partitioned into contiguous ranges, each rank
runs stage 1 on its range, the per-window logits (N x 2 fp32 — a few KB) are all-gathered over RCCL/xGMI, every rank
derives the identical gate, the K gated windows are re-partitioned evenly (swallows cluster in time, so re-using the
stage-1 partition would imbalance stage 2) and the stage-2 logits are all-gathered the same way.  No other data-path
collective exists.

The GPU path gathers through the C ABI (`zk_allgather_logits`: RCCL inside libzkast.so on the context's stream; the
host only ships the 128-byte unique id once, see `init_comm`).  The compute and the gather are injected as callables so
that the partition / gather / gate logic is testable on CPU (`torch.distributed` gloo, world_size 2) without a GPU;
`ZkShardedCascade` binds them to the HIP path.
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


def init_comm(ctx, rank: int, world: int, exchange=None) -> None:
    """Bind an RCCL communicator to the context.  `exchange(payload_or_None) -> bytes` ships rank 0's 128-byte unique
    id to every rank over a host channel; default: a broadcast on the already initialised torch.distributed group (the
    only thing torch.distributed is used for on the GPU path)."""
    from . import lib
    if world == 1:
        ctx.comm_init(0, 1, None)
        return
    import os
    if os.environ.get("MASTER_ADDR", "127.0.0.1") in ("127.0.0.1", "localhost"):
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")      # one node: bootstrap over loopback (the host name may not resolve)
    uid = lib.comm_unique_id() if rank == 0 else None
    if exchange is None:
        def exchange(payload):
            box = [payload]
            _dist().broadcast_object_list(box, src=0)
            return box[0]
    ctx.comm_init(rank, world, exchange(uid))


def all_gather_rows(local: np.ndarray, n_total: int, world: int, device=None, ctx=None) -> np.ndarray:
    """all-gather of contiguous row shards produced with shard_range; returns (n_total, cols) on every rank.
    Shards are padded to the largest count so that ONE fixed-size collective is issued.  ctx: gather through the C ABI
    (RCCL in libzkast.so); otherwise `torch.distributed` (gloo in the CPU tests)."""
    cols = local.shape[1] if local.ndim == 2 else 2
    per = (n_total + world - 1) // world
    if ctx is not None and world > 1:
        buf = np.zeros((per, cols), np.float32)
        if local.shape[0]:
            buf[: local.shape[0]] = np.ascontiguousarray(local, dtype=np.float32)
        out = np.empty((world, per, cols), np.float32)
        ctx.allgather_logits(buf, per, cols, out)
    else:
        import torch
        dist = _dist()
        if world == 1 or not (dist.is_available() and dist.is_initialized()):
            return np.ascontiguousarray(local, dtype=np.float32).reshape(n_total, cols)
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
                    softmax: Callable[[np.ndarray], np.ndarray] = softmax_np, ctx=None, stats: Optional[dict] = None):
    """stage_logits(stage, win_idx int32[]) -> (len(win_idx), 2) logits of those windows, computed locally.
    Returns (s1_logits (N,2), swallow_idx (K,), s2_logits (K,2)) — identical on every rank.
    stats (optional dict): "gather_s" += wall time of the two logit all-gathers (includes waiting for the slowest rank),
    "gathers" += their count."""
    import time

    def gather(local, total):
        t0 = time.perf_counter()
        out = all_gather_rows(local, total, world, device, ctx)
        if stats is not None:
            stats["gather_s"] = stats.get("gather_s", 0.0) + (time.perf_counter() - t0)
            stats["gathers"] = stats.get("gathers", 0) + 1
        return out

    lo, hi = shard_range(n_windows, rank, world)
    mine = np.arange(lo, hi, dtype=np.int32)
    l1 = stage_logits(0, mine) if hi > lo else np.zeros((0, 2), np.float32)
    s1 = gather(l1, n_windows)
    idx = gate_indices(softmax(s1) if n_windows else np.zeros((0, 2), np.float32), thr1, fwd_min_prob)
    k = int(idx.shape[0])
    klo, khi = shard_range(k, rank, world)
    l2 = stage_logits(1, idx[klo:khi]) if khi > klo else np.zeros((0, 2), np.float32)
    s2 = gather(l2, k) if k else np.zeros((0, 2), np.float32)
    return s1, idx, s2


class WavSource:
    """A recording that is still file bytes (RIFF data chunk + format): a rank decodes and resamples ON THE DEVICE only
    the frames its windows need (BASELINE configs[3]: a 30-min 48 kHz file is 173 MB of PCM16, of which a rank of 8
    uploads an eighth).  Slices start on a multiple of the resampler's input period, so an output sample of a slice is
    the same float as in the whole-recording result; a margin wider than the sinc kernel keeps every tap inside."""

    def __init__(self, raw: bytes, format_tag: int, bits: int, channels: int, sr: int, target_sr: int = 16000):
        self.raw, self.tag, self.bits, self.ch, self.sr, self.target = raw, format_tag, bits, channels, sr, target_sr
        self.frame_bytes = channels * (bits // 8)
        self.n_frames = len(raw) // self.frame_bytes
        g = int(np.gcd(sr, target_sr))
        self.orig, self.neu = sr // g, target_sr // g
        self.n_samples = (self.neu * self.n_frames + self.orig - 1) // self.orig if sr != target_sr else self.n_frames
        width = int(np.ceil(6.0 * self.orig / (min(self.orig, self.neu) * 0.99)))
        self.margin = max(2, (width + self.orig) // self.orig + 1)      # in blocks of `orig` input frames

    @classmethod
    def from_file(cls, path: str, target_sr: int = 16000):
        from .pipeline import parse_wav
        tag, ch, sr, bits, raw = parse_wav(path)
        return cls(raw, tag, bits, ch, sr, target_sr)

    def load(self, ctx, a0: int, a1: int):
        """leave output samples covering [a0, a1) in the context's audio slot; returns (offset of a0 in the slot, bytes
        uploaded)"""
        q0 = max(0, a0 // self.neu - self.margin)
        q1 = -(-a1 // self.neu) + self.margin
        f0, f1 = q0 * self.orig, min(self.n_frames, q1 * self.orig)
        piece = self.raw[f0 * self.frame_bytes: f1 * self.frame_bytes]
        ctx.audio_load(piece, self.tag, self.bits, self.ch, self.sr, self.target)
        return a0 - q0 * self.neu, len(piece)


class ZkShardedCascade:
    """Binds sharded_cascade to the HIP path for one recording.  A rank uploads and log-mels only the audio its windows
    need: `audio[lo*hop : (hi-1)*hop + win]` for its stage-1 range [lo, hi) (adjacent ranks overlap by
    win - hop samples), and, when its share of the re-partitioned gated windows leaves that range, the slice that
    covers the share.  `audio` is a 16 kHz float array or a `WavSource` (file bytes: decoded and resampled on the
    device, slice by slice).  The logit gathers run through `zk_allgather_logits` when the context has a communicator
    (`init_comm`), else through `torch.distributed`."""

    def __init__(self, model_s1, fx_s1, model_s2, fx_s2, rank: int, world: int, device=None, use_ctx_comm=None):
        self.m = (model_s1, model_s2)
        self.fx = (fx_s1, fx_s2)
        self.rank, self.world, self.device = rank, world, device
        ctx = model_s1._ctx
        if use_ctx_comm is None:
            use_ctx_comm = ctx.comm_info()[1] == world and world > 1
        self.comm_ctx = ctx if use_ctx_comm else None
        self.h2d_samples = 0          # audio samples (array) or bytes (WavSource) uploaded by the last call
        self.uploads = 0              # slices uploaded by the last call (1 = stage 2 stayed inside the stage-1 slice)
        self.stats = {}               # gather_s / gathers of the last call (sharded_cascade)
        self.n_windows = 0

    def __call__(self, audio, window_sec=1.0, hop_sec=0.5, thr1=0.5, fwd_min_prob=None):
        from .pipeline import window_geometry
        src = audio if isinstance(audio, WavSource) else None
        if src is None:
            audio = np.ascontiguousarray(audio, dtype=np.float32)
        n_samples = src.n_samples if src is not None else len(audio)
        n, win, hop = window_geometry(n_samples, window_sec, hop_sec)
        self.n_windows = n
        ctx = self.m[0]._ctx
        for m, fx in zip(self.m, self.fx):
            m.bind_feature_extractor(fx)
        self.h2d_samples = 0
        self.uploads = 0
        self.stats = {}
        slot = [0, 0]                 # window range [lo, hi) currently in the feature slot

        def fill_slot(lo, hi):
            if lo >= slot[0] and hi <= slot[1]:
                return
            self.uploads += 1
            a0 = lo * hop
            a1 = min(n_samples, (hi - 1) * hop + win)      # a recording shorter than one window is zero-padded
            if src is not None:
                off, nbytes = src.load(ctx, a0, a1)
                ctx.logmel(None, 0, off, hop, win, hi - lo)
                self.h2d_samples += nbytes
            else:
                piece = audio[a0:a1]
                ctx.logmel(piece, len(piece), 0, hop, win, hi - lo)
                self.h2d_samples += len(piece)
            slot[0], slot[1] = lo, hi

        def stage_logits(stage, win_idx):
            win_idx = np.asarray(win_idx, np.int32)
            fill_slot(int(win_idx[0]), int(win_idx[-1]) + 1)
            return self.m[stage].forward_from_slot(len(win_idx), win_idx - np.int32(slot[0]))

        return sharded_cascade(n, stage_logits, self.rank, self.world, thr1, fwd_min_prob, self.device, ctx.softmax,
                               self.comm_ctx, self.stats)

"""Feature cache for the cached variant of the two-stage script (src/test_long_audio_windows_2stage_cache.py:84-208).

The reference stores, per recording and extractor, a torch bundle of the PADDED and NORMALISED extractor output:
(N, 1024, 128) float32 — 512 KiB per window, of which 926 of the 1024 rows are the constant pad value, and one bundle
per (mean, std) pair because the normalisation is baked in.  This build keeps what the extractor actually computes:

    CompactFeatures   (N, n_frames, 128) float32 UN-normalised log-mel (n_frames = 98 for 1-s windows: 49 KiB per
                      window, 10.4x smaller), the same array the library holds on the device ("feature slot").
                      Normalisation is an affine map applied by whoever reads the slot, so ONE store serves both
                      stages whatever their mean/std (the reference re-uses stage-1 features for stage 2 only when the
                      two extractors are identical, ..._cache.py:418-422).
    FeatureCache      a directory of such stores (`<stem>_<digest>.zkc.npz`) that ALSO reads and writes the reference's
                      `<stem>_<digest>.pt` bundles, byte-compatible in key, digest and metadata, so caches made by
                      either side are interchangeable.

The byte formats that make the caches interchangeable — extractor fingerprint, cache key string, bundle keys and
metadata fields — are the contract and are reproduced exactly (tests/test_host_logic.py pins them against values
computed with the real ASTFeatureExtractor); everything else here is this build's own design.  The functions at the end
keep the cached script's call signatures on top of the two classes.
"""
from __future__ import annotations

import hashlib
import json
import os
from dataclasses import dataclass, field
from typing import Any, Dict, Optional, Sequence

import numpy as np

from . import lib as _lib

SAMPLING_RATE = 16000
N_MEL = 128
MAX_FRAMES = 1024
_COMPACT_MAGIC = "zkast-compact-logmel-v1"


# ------------------------------------------------------------------------------------------------------------------
# identity of a cache entry (contract with the reference: ..._cache.py:84-124)
# ------------------------------------------------------------------------------------------------------------------
def get_fx_fingerprint(fx) -> str:
    """sha256 over the extractor's sorted-key JSON (:84-86)."""
    blob = json.dumps(fx.to_dict(), sort_keys=True)
    return hashlib.sha256(blob.encode("utf-8")).hexdigest()


@dataclass(frozen=True)
class EntryKey:
    """What identifies one cached recording: the file (path, size, mtime), the window grid and the extractor."""
    audio_abs: str
    size: int
    mtime: int
    window_sec: float
    hop_sec: float
    sr: int
    fingerprint: str

    @classmethod
    def of(cls, audio_path: str, window_sec: float, hop_sec: float, sr: int, fingerprint: str) -> "EntryKey":
        p = os.path.abspath(audio_path)
        st = os.stat(p)
        return cls(p, st.st_size, int(st.st_mtime), window_sec, hop_sec, sr, fingerprint)

    @property
    def digest(self) -> str:
        fields = (self.audio_abs, self.window_sec, self.hop_sec, self.sr, self.fingerprint, f"{self.size}_{self.mtime}")
        return hashlib.sha256("|".join(str(v) for v in fields).encode("utf-8")).hexdigest()[:16]

    @property
    def stem(self) -> str:
        return f"{os.path.splitext(os.path.basename(self.audio_abs))[0]}_{self.digest}"

    def metadata(self, num_windows: int) -> Dict[str, Any]:
        return dict(audio_path=self.audio_abs, audio_size=self.size, audio_mtime=self.mtime, window_sec=self.window_sec,
                    hop_sec=self.hop_sec, num_windows=num_windows, sampling_rate=self.sr,
                    extractor_fingerprint=self.fingerprint)


def build_cache_path(cache_dir: str, audio_path: str, window_sec: float, hop_sec: float, sr: int,
                     fx_fingerprint: str) -> str:
    """Path of the reference-format bundle of this entry (:89-103)."""
    return os.path.join(cache_dir, EntryKey.of(audio_path, window_sec, hop_sec, sr, fx_fingerprint).stem + ".pt")


def build_base_metadata(audio_path: str, window_sec: float, hop_sec: float, num_windows: int, sr: int,
                        fx_fingerprint: str) -> Dict[str, Any]:
    """The metadata block a bundle must match to be accepted (:106-124)."""
    return EntryKey.of(audio_path, window_sec, hop_sec, sr, fx_fingerprint).metadata(num_windows)


# ------------------------------------------------------------------------------------------------------------------
# the compact store
# ------------------------------------------------------------------------------------------------------------------
def _norm_params(fx):
    """(mean, 2*std) of the extractor's normalize(), or None when it does not normalise."""
    return (float(fx.mean), 2.0 * float(fx.std)) if getattr(fx, "do_normalize", True) else None


@dataclass
class CompactFeatures:
    """Un-normalised log-mel of N windows, host side: logmel (N, n_frames, 128) float32."""
    logmel: np.ndarray
    provenance: Dict[str, Any] = field(default_factory=dict)

    def __post_init__(self):
        a = np.ascontiguousarray(self.logmel, dtype=np.float32)
        if a.ndim != 3 or a.shape[2] != N_MEL or not 1 <= a.shape[1] <= MAX_FRAMES:
            raise ValueError(f"compact features must be (N, 1..{MAX_FRAMES}, {N_MEL}), got {a.shape}")
        self.logmel = a

    def __len__(self) -> int:
        return self.logmel.shape[0]

    @property
    def n_frames(self) -> int:
        return self.logmel.shape[1]

    # -- device slot ------------------------------------------------------------------------------------------
    @classmethod
    def from_device(cls, ctx, **provenance) -> "CompactFeatures":
        """Copy of the library's feature slot (what the last zk_logmel / zk_two_stage left there)."""
        return cls(ctx.features_get(), dict(provenance))

    def to_device(self, ctx) -> None:
        """Make this store the library's feature slot; forwards then read it like fresh zk_logmel output."""
        ctx.features_set(self.logmel)

    # -- the reference's padded / normalised form -------------------------------------------------------------
    def expand(self, fx) -> np.ndarray:
        """(N, max_length, 128) float32 as ASTFeatureExtractor.__call__ returns it: zero rows appended up to
        max_length, then (x - mean) / (2 std) over ALL rows (pad rows become -mean / (2 std))."""
        rows = int(getattr(fx, "max_length", MAX_FRAMES))
        out = np.zeros((len(self), rows, N_MEL), np.float32)
        out[:, : min(rows, self.n_frames)] = self.logmel[:, :rows]
        nrm = _norm_params(fx)
        if nrm is not None:
            out -= np.float32(nrm[0])
            out /= np.float32(nrm[1])
        return out

    @classmethod
    def from_expanded(cls, features, fx, n_frames: int, **provenance) -> "CompactFeatures":
        """Inverse of expand() for a reference bundle: drop the pad rows, undo the affine normalisation in float64 and
        round once (the round trip differs from the extractor's own output by <= 1 ulp of the log-mel value)."""
        x = np.asarray(features, dtype=np.float32)
        if x.ndim != 3 or x.shape[2] != N_MEL or x.shape[1] < n_frames:
            raise ValueError(f"expanded features must be (N, >={n_frames}, {N_MEL}), got {x.shape}")
        real = x[:, :n_frames].astype(np.float64)
        nrm = _norm_params(fx)
        if nrm is not None:
            real = real * np.float64(np.float32(nrm[1])) + np.float64(np.float32(nrm[0]))
        return cls(real.astype(np.float32), dict(provenance))

    # -- own on-disk format -----------------------------------------------------------------------------------
    def save(self, path: str, metadata: Optional[Dict[str, Any]] = None) -> None:
        head = dict(magic=_COMPACT_MAGIC, shape=list(self.logmel.shape), metadata=metadata or {},
                    provenance=self.provenance)
        tmp = path + ".part"
        try:
            with open(tmp, "wb") as f:
                np.savez(f, header=np.frombuffer(json.dumps(head).encode("utf-8"), np.uint8), logmel=self.logmel)
            os.replace(tmp, path)      # readers never see a half-written store
        except BaseException:
            try:                       # a failed write (disk full, interrupt) leaves no stray .part file behind
                os.unlink(tmp)
            except OSError:
                pass
            raise

    @classmethod
    def load(cls, path: str):
        """-> (CompactFeatures, metadata dict).  Raises ValueError on anything that is not a store of this format."""
        with np.load(path, allow_pickle=False) as z:
            if "header" not in z.files or "logmel" not in z.files:
                raise ValueError(f"{path}: not a compact feature store")
            head = json.loads(bytes(z["header"]).decode("utf-8"))
            arr = z["logmel"]
        if head.get("magic") != _COMPACT_MAGIC or list(arr.shape) != head.get("shape"):
            raise ValueError(f"{path}: header does not describe the stored array")
        return cls(arr, head.get("provenance", {})), head.get("metadata", {})


def n_frames_of(window_sec: float, sr: int = SAMPLING_RATE) -> int:
    """Real (un-padded) frames the extractor produces for one window: 1 + (win - 400) // 160, capped at max_length."""
    win = int(window_sec * sr)
    return max(1, min(MAX_FRAMES, 1 + (win - 400) // 160))


# ------------------------------------------------------------------------------------------------------------------
# the cache directory
# ------------------------------------------------------------------------------------------------------------------
class FeatureCache:
    """Directory of cached recordings.  Lookup order: this build's compact store, then the reference's `.pt` bundle of
    the same key (imported and compacted).  `write_reference_bundle=True` also emits the `.pt` twin on store(), which
    is what keeps a cache directory usable by the reference script."""

    def __init__(self, directory: str, write_reference_bundle: bool = True, log=print):
        self.directory = directory
        self.write_reference_bundle = write_reference_bundle
        self.log = log
        os.makedirs(directory, exist_ok=True)

    def _paths(self, key: EntryKey):
        base = os.path.join(self.directory, key.stem)
        return base + ".zkc.npz", base + ".pt"

    @staticmethod
    def _accepts(found: Dict[str, Any], wanted: Dict[str, Any]) -> bool:
        return all(found.get(k) == v for k, v in wanted.items())

    def lookup(self, key: EntryKey, num_windows: int, fx, tag: str = "stage1") -> Optional[CompactFeatures]:
        wanted = key.metadata(num_windows)
        compact_path, bundle_path = self._paths(key)
        for path, reader in ((compact_path, self._read_compact), (bundle_path, self._read_bundle)):
            if not os.path.exists(path):
                continue
            try:
                feats, meta = reader(path, fx, key)
            except Exception as exc:      # unreadable entry: report, fall through to the next source / recompute
                self.log(f"[cache:{tag}] cannot use {path} ({type(exc).__name__}: {exc})")
                continue
            if self._accepts(meta, wanted) and len(feats) == num_windows:
                self.log(f"[cache:{tag}] Loaded {path}")
                return feats
            self.log(f"[cache:{tag}] {path} describes another recording / window grid; ignored")
        return None

    @staticmethod
    def _read_compact(path, _fx, _key):
        return CompactFeatures.load(path)

    @staticmethod
    def _read_bundle(path, fx, key):
        import torch
        bundle = torch.load(path, map_location="cpu")
        meta = dict(bundle.get("metadata", {}))
        feats = CompactFeatures.from_expanded(bundle["features"].numpy(), fx, n_frames_of(key.window_sec, key.sr),
                                              imported_from=os.path.basename(path))
        return feats, meta

    def store(self, key: EntryKey, feats: CompactFeatures, fx, tag: str = "stage1") -> None:
        """Best effort, like the reference (…_cache.py:181-192 catches Exception and carries on): a read-only or full cache
        directory, a failing torch.save (RuntimeError) or the (N,1024,128) expansion of the `.pt` twin running out of memory
        (MemoryError; 512 KiB per window) must not stop the patient's inference.  The two files are written independently:
        a failing twin does not take the compact store with it."""
        meta = key.metadata(len(feats))
        compact_path, bundle_path = self._paths(key)
        try:
            feats.save(compact_path, meta)
            self.log(f"[cache:{tag}] Saved {compact_path}")
        except Exception as exc:          # noqa: BLE001
            self.log(f"[cache:{tag}] could not write {compact_path}: {type(exc).__name__}: {exc}")
        if not self.write_reference_bundle:
            return
        tmp = bundle_path + ".part"
        try:
            write_reference_bundle(tmp, feats.expand(fx), meta)
            os.replace(tmp, bundle_path)
            self.log(f"[cache:{tag}] Saved {bundle_path}")
        except Exception as exc:          # noqa: BLE001
            self.log(f"[cache:{tag}] could not write {bundle_path}: {type(exc).__name__}: {exc}")
            try:
                os.unlink(tmp)
            except OSError:
                pass


def write_reference_bundle(path: str, expanded: np.ndarray, base_metadata: Dict[str, Any]) -> None:
    """torch.save of {"metadata": {..., "feature_shape"}, "features": FloatTensor (N,1024,128)} (:181-187)."""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(expanded, dtype=np.float32))
    torch.save({"metadata": {**base_metadata, "feature_shape": list(t.shape)}, "features": t}, path)


# ------------------------------------------------------------------------------------------------------------------
# the cached script's entry points on top of the two classes
# ------------------------------------------------------------------------------------------------------------------
def extract_compact(fx, windows: Sequence[np.ndarray], device: int = 0) -> CompactFeatures:
    """Log-mel of a list of equally long windows on the GPU, kept compact: the windows are laid end to end and cut
    again by the kernel (hop = window length), one launch for the whole list."""
    if len(windows) == 0:
        raise RuntimeError("no windows to extract features from (empty recording or window grid)")
    win = len(windows[0])
    if any(len(w) != win for w in windows):
        raise ValueError("windows of one recording must all have the same length")
    ctx = _lib.get_context(getattr(fx, "_device", device))
    flat = np.concatenate([np.asarray(w, dtype=np.float32) for w in windows])
    ctx.logmel(flat, flat.shape[0], 0, win, win, len(windows))
    return CompactFeatures.from_device(ctx, extractor="zk_logmel")


def compute_features(fx, windows: Sequence[np.ndarray], batch_size: int = 0):
    """(N, 1024, 128) float32 CPU tensor, the cached script's feature tensor (:127-139).  `batch_size` is accepted for
    signature compatibility; the device extractor takes the whole list at once."""
    import torch
    return torch.from_numpy(extract_compact(fx, windows).expand(fx))


def load_or_compute_features(audio_path: str, windows: Sequence[np.ndarray], fx, window_sec: float, hop_sec: float,
                             batch_size: int, cache_dir: Optional[str], disable_cache: bool = False,
                             refresh_cache: bool = False, stage_label: str = "stage1", log=print, compact: bool = False):
    """The cached script's feature step (:142-192) over FeatureCache.  Returns the (N,1024,128) tensor the script works
    with, or the CompactFeatures themselves with compact=True (what classify-from-cache uses: no 10x expansion)."""
    import torch
    feats = None
    cache = key = None
    if cache_dir and not disable_cache:
        cache = FeatureCache(cache_dir, log=log)
        key = EntryKey.of(audio_path, window_sec, hop_sec, SAMPLING_RATE, get_fx_fingerprint(fx))
        if not refresh_cache:
            feats = cache.lookup(key, len(windows), fx, stage_label)
    else:
        log(f"[cache:{stage_label}] cache off: extracting features")
    if feats is None:
        feats = extract_compact(fx, windows)
        if cache is not None:
            cache.store(key, feats, fx, stage_label)
    return feats if compact else torch.from_numpy(feats.expand(fx))


def forward_probs_from_features(model, features, batch_size: int) -> np.ndarray:
    """Softmax probabilities (N, labels) float32 from cached features (:198-208); no rows -> np.zeros((0, 0)).
    `features` is either the script's (N,1024,128) tensor / array (forwarded in slices of batch_size, as the script
    does) or a CompactFeatures store, which goes to the device slot once and is normalised there with the mean/std of
    the extractor bound to `model` — the affine stage-2 re-use."""
    ctx = _lib.get_context(getattr(model, "_device", 0))
    if isinstance(features, CompactFeatures):
        if len(features) == 0:
            return np.zeros((0, 0))
        features.to_device(ctx)
        return ctx.softmax(model.forward_from_slot(len(features)))
    total = int(features.shape[0])
    if total == 0:
        return np.zeros((0, 0))
    out = []
    for lo in range(0, total, max(1, int(batch_size))):
        logits = np.asarray(model(features[lo:lo + batch_size]).logits)
        out.append(ctx.softmax(logits))
    return np.concatenate(out, axis=0)

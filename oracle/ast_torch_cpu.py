"""CPU BASELINE — TEST INFRASTRUCTURE ONLY (same rules as ast_oracle.py: imported by tests/ and by bench.py's
``cpu_baseline`` leg, never by the product path).

The reference's hot path on host cores, the way the reference itself runs it on a CPU: float32 torch operators
(``F.linear``, ``F.scaled_dot_product_attention``, ``F.gelu``, ``F.layer_norm`` — what
``ASTForAudioClassification`` dispatches to, ``$TF/models/audio_spectrogram_transformer/modeling_audio_spectrogram_transformer.py:57-61,
102-127,145-176,187-224,302-318``) behind the numpy float64 log-mel of ``ASTFeatureExtractor``'s numpy branch
(``ast_oracle.fbank_frames``, bit-exact against the golden fixture).  This is a restatement, not the reference (``kind:
"port"`` in the bench line): ``tests/test_oracle.py::test_torch_cpu_baseline_matches_golden`` pins its logits against
``tests/golden/model.npz`` (real transformers fp32) to <= 1e-4.

SURVEY.md §8d asks for N = 64 windows, 3 repeats, median, all host cores, thread count stated.  ``effective_cpus()``
is the number of cores this process may actually use (affinity mask AND cgroup quota): asking torch / BLAS for more
threads than that — e.g. all 256 hardware threads of a GPU box on which the job owns 16 — thrashes instead of scaling.
"""
from __future__ import annotations

import os
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import ast_oracle as orc


def effective_cpus() -> int:
    """min(affinity mask, cgroup cpu quota), at least 1."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:                                         # cgroup v2
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:                                     # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and p > 0:
                n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return max(1, n)


def extract_features_parallel(windows, mean: float, std: float, threads: int) -> np.ndarray:
    """ast_oracle.extract_features with the per-window log-mel spread over a thread pool (numpy's FFT and matmul
    release the GIL); same arithmetic per window, so the result is bit-identical."""
    out = np.zeros((len(windows), orc.MAX_LEN, orc.N_MEL), dtype=np.float32)

    def one(i):
        fb = orc.fbank_frames(np.asarray(windows[i], dtype=np.float32))
        out[i, : min(fb.shape[0], orc.MAX_LEN)] = fb[: orc.MAX_LEN]

    with ThreadPoolExecutor(max_workers=max(1, threads)) as pool:
        list(pool.map(one, range(len(windows))))
    return ((out - np.float32(mean)) / np.float32(std * 2)).astype(np.float32)


class TorchAST:
    """AST forward on torch-CPU float32 operators from a numpy state dict (either transformers key scheme)."""

    def __init__(self, sd):
        import torch
        self.torch = torch
        W = sd if isinstance(sd, orc.ASTWeights) else orc.ASTWeights(sd)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))      # noqa: E731
        p = W.p + "embeddings."
        self.conv_w = t(W.g(p + "patch_embeddings.projection.weight"))                 # (768,1,16,16)
        self.conv_b = t(W.g(p + "patch_embeddings.projection.bias"))
        self.cls, self.dist = t(W.g(p + "cls_token")), t(W.g(p + "distillation_token"))
        self.pos = t(W.g(p + "position_embeddings"))
        self.layers = []
        for i in range(W.n_layers):
            L = W.layer(i)
            qkv_w = np.concatenate([L["q"][0], L["k"][0], L["v"][0]], 0)
            qkv_b = np.concatenate([L["q"][1], L["k"][1], L["v"][1]], 0)
            self.layers.append(dict(ln1=(t(L["ln1"][0]), t(L["ln1"][1])), ln2=(t(L["ln2"][0]), t(L["ln2"][1])),
                                    qkv=(t(qkv_w), t(qkv_b)), o=(t(L["o"][0]), t(L["o"][1])),
                                    fc1=(t(L["fc1"][0]), t(L["fc1"][1])), fc2=(t(L["fc2"][0]), t(L["fc2"][1]))))
        self.lnf = (t(W.g(W.p + "layernorm.weight")), t(W.g(W.p + "layernorm.bias")))
        self.lnh = (t(W.g("classifier.layernorm.weight")), t(W.g("classifier.layernorm.bias")))
        self.head = (t(W.g("classifier.dense.weight")), t(W.g("classifier.dense.bias")))

    def forward(self, input_values: np.ndarray, chunk: int = 8) -> np.ndarray:
        torch = self.torch
        F = torch.nn.functional
        outs = []
        with torch.inference_mode():
            x_all = torch.from_numpy(np.ascontiguousarray(input_values, dtype=np.float32))
            for b0 in range(0, x_all.shape[0], chunk):
                x = x_all[b0:b0 + chunk]
                B = x.shape[0]
                # (B,1024,128) -> (B,1,128 freq,1024 time) -> Conv2d(1,768,16,stride 10) -> (B,1212,768), token = f*101 + t
                h = F.conv2d(x.unsqueeze(1).transpose(2, 3), self.conv_w, self.conv_b, stride=(orc.FSTRIDE, orc.TSTRIDE))
                h = h.flatten(2).transpose(1, 2)
                h = torch.cat([self.cls.expand(B, -1, -1), self.dist.expand(B, -1, -1), h], dim=1) + self.pos
                for L in self.layers:
                    y = F.layer_norm(h, (orc.HIDDEN,), L["ln1"][0], L["ln1"][1], orc.LN_EPS)
                    qkv = F.linear(y, *L["qkv"]).view(B, orc.SEQ, 3, orc.HEADS, orc.HEAD_DIM).permute(2, 0, 3, 1, 4)
                    a = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2])
                    h = h + F.linear(a.transpose(1, 2).reshape(B, orc.SEQ, orc.HIDDEN), *L["o"])
                    y = F.layer_norm(h, (orc.HIDDEN,), L["ln2"][0], L["ln2"][1], orc.LN_EPS)
                    h = h + F.linear(F.gelu(F.linear(y, *L["fc1"])), *L["fc2"])
                seq = F.layer_norm(h[:, :2], (orc.HIDDEN,), self.lnf[0], self.lnf[1], orc.LN_EPS)
                pooled = (seq[:, 0] + seq[:, 1]) / 2
                z = F.layer_norm(pooled, (orc.HIDDEN,), self.lnh[0], self.lnh[1], orc.LN_EPS)
                outs.append(F.linear(z, *self.head))
        return torch.cat(outs, 0).numpy() if outs else np.zeros((0, 2), np.float32)


def time_two_stage(windows, sd1, sd2, fx1, fx2, repeats: int = 3, threads: int | None = None,
                   budget_s: float = 150.0, mel_threads: int | None = None):
    """Median wall time of (log-mel + forward) x 2 stages over `windows`, every window through both stages (the bench's
    g = 1.0 workload).  Stops repeating early rather than exceed `budget_s` (the bench line says how many repeats ran).
    `mel_threads` = size of the log-mel thread pool (default: `threads`; 1 = no pool concurrency, bench.py's retry).
    -> dict(seconds, mel_seconds, forward_seconds, repeats, threads, logits1, logits2)."""
    import torch
    threads = threads or effective_cpus()
    mel_threads = mel_threads or threads
    prev = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        m1, m2 = TorchAST(sd1), TorchAST(sd2)
        runs, t_start, l1, l2 = [], time.perf_counter(), None, None
        for _ in range(max(1, repeats)):
            t0 = time.perf_counter()
            f1 = extract_features_parallel(windows, fx1[0], fx1[1], mel_threads)
            t1 = time.perf_counter()
            l1 = m1.forward(f1)
            t2 = time.perf_counter()
            f2 = extract_features_parallel(windows, fx2[0], fx2[1], mel_threads)
            t3 = time.perf_counter()
            l2 = m2.forward(f2)
            t4 = time.perf_counter()
            runs.append((t4 - t0, (t1 - t0) + (t3 - t2), (t2 - t1) + (t4 - t3)))
            if (time.perf_counter() - t_start) + runs[-1][0] > budget_s:
                break
        runs.sort()
        tot, mel, fwd = runs[len(runs) // 2]
        return dict(seconds=tot, mel_seconds=mel, forward_seconds=fwd, repeats=len(runs), threads=threads, logits1=l1,
                    logits2=l2)
    finally:
        torch.set_num_threads(prev)

"""The bench's own launch shape on a real MI355X (BASELINE configs[2] and the single-rank shape of configs[3]):
B = 1024 windows through `classify_recording` (the library picks 2 micro-batches of 512 windows = 621,568 token rows per
GEMM launch) and one 3,599-window recording (30 min at 16 kHz: 8 micro-batches of 450).  At these sizes the oracle
cannot run the whole batch, so the checks are the ones that do not depend on size — bit-invariance against the
micro-batch split, against a permutation of the window order and against running a window range on its own — plus an
oracle spot-check of windows spread over both micro-batches and both stages.
Reference path: src/test_long_audio_windows_2stage.py:301-348."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import ast_oracle as orc  # noqa: E402  (checker only)

S1 = (-1.1509622, 3.5340312)
S2 = (-6.5, 2.75)
HOP, WIN = 8000, 16000


def _load(stage, sd, stats):
    from zkast import ZkASTConfig, ZkASTFeatureExtractor, ZkASTForAudioClassification
    m = ZkASTForAudioClassification(ZkASTConfig(num_labels=2), sd, stage=stage, fx_mean=stats[0], fx_std=stats[1])
    return m, ZkASTFeatureExtractor(mean=stats[0], std=stats[1])


def _shifted(sd, shift):
    out = dict(sd)
    out["classifier.dense.bias"] = sd["classifier.dense.bias"].copy()
    out["classifier.dense.bias"][1] += np.float32(shift)
    return out


@pytest.fixture(scope="module")
def cascade():
    """both stages resident; the stage-1 swallow bias is shifted to the median logit margin of the 1024-window batch so
    that about half of the windows pass the gate (random weights would otherwise gate arbitrarily)"""
    from zkast import lib, synth
    ctx = lib.get_context(0)
    ctx.set_micro_batch(0)
    sd1, sd2 = synth.make_ast_weights(21, "wide"), synth.make_ast_weights(22, "wide")
    B = 1024
    rec = synth.synth_recording(100, WIN + (B - 1) * HOP)
    m1, fx1 = _load(0, sd1, S1)
    ctx.logmel(rec, rec.size, 0, HOP, WIN, B)
    l = m1.forward_from_slot(B)
    shift = float(-np.median(l[:, 1] - l[:, 0]))
    sd1s = _shifted(sd1, shift)
    m1, fx1 = _load(0, sd1s, S1)
    m2, fx2 = _load(1, sd2, S2)
    return dict(ctx=ctx, rec=rec, B=B, m1=m1, fx1=fx1, m2=m2, fx2=fx2, sd1=sd1s, sd2=sd2)


def test_b1024_cascade_at_the_bench_launch_shape(cascade):
    from zkast import pipeline as pl
    c = cascade
    ctx, rec, B = c["ctx"], c["rec"], c["B"]
    ctx.set_micro_batch(0)                                           # auto: 2 x 512 windows
    summ, p1, preds, aligned, res2 = pl.classify_recording(rec, c["m1"], c["fx1"], c["m2"], c["fx2"])
    idx = np.array([g for g, _ in res2], np.int64)
    p2 = np.stack([p for _, p in res2])
    assert summ["num_windows"] == B and p1.shape == (B, 2) and 0.3 * B < len(idx) < 0.7 * B
    assert np.array_equal(idx, np.where((p1.argmax(1) == 1) & (p1[:, 1] >= 0.5))[0])
    assert (idx < 512).any() and (idx >= 512).any()                  # the gate picks from both micro-batches
    assert np.all(np.isfinite(p1)) and np.all(np.isfinite(p2)) and np.allclose(p1.sum(1), 1.0, atol=1e-6)
    # --- micro-batch split invariance: a window's arithmetic does not depend on what shares its launch ---
    for mb in (256, 107):
        ctx.set_micro_batch(mb)
        _s, q1, _p, _a, r2 = pl.classify_recording(rec, c["m1"], c["fx1"], c["m2"], c["fx2"])
        assert np.array_equal(q1, p1), mb
        assert [g for g, _ in r2] == idx.tolist() and np.array_equal(np.stack([p for _, p in r2]), p2), mb
    ctx.set_micro_batch(0)
    # --- the device-resident recording (audio slot) gives the same bits as the host array ---
    # --- permutation invariance: stage-1 logits of a permuted window list are the permuted logits ---
    ctx.logmel(rec, rec.size, 0, HOP, WIN, B)
    l_id = c["m1"].forward_from_slot(B)
    perm = np.random.default_rng(0).permutation(B).astype(np.int32)
    l_pm = c["m1"].forward_from_slot(B, perm)
    assert np.array_equal(l_pm, l_id[perm])
    assert np.array_equal(ctx.softmax(l_id), p1)
    # --- oracle spot-check: 8 windows over both micro-batches, stage 1; 4 gated windows, stage 2 ---
    wins = orc.window_audio(rec)
    pick1 = [0, 255, 511, 512, 513, 777, 1000, 1023]
    W1, W2 = orc.ASTWeights(c["sd1"]), orc.ASTWeights(c["sd2"])
    ref1 = orc.softmax(orc.ast_forward(orc.extract_features([wins[i] for i in pick1], *S1), W1))
    assert np.abs(p1[pick1] - ref1).max() <= 5e-4
    pick2 = [int(idx[0]), int(idx[len(idx) // 2 - 1]), int(idx[len(idx) // 2]), int(idx[-1])]
    ref2 = orc.softmax(orc.ast_forward(orc.extract_features([wins[i] for i in pick2], *S2), W2))
    got2 = np.stack([p2[np.searchsorted(idx, i)] for i in pick2])
    assert np.abs(got2 - ref2).max() <= 5e-4


def test_3599_window_recording_single_rank_shape(cascade):
    """configs[3]'s per-recording shape on one rank: 28.8 M samples -> 3599 windows -> 8 micro-batches of 450; the
    recording goes up once and stays in the audio slot (48 kHz PCM16 file bytes -> device decode + resample)."""
    import struct
    from zkast import pipeline as pl, synth
    c = cascade
    ctx = c["ctx"]
    n48 = 30 * 60 * 48000
    x48 = synth.synth_recording(7, n48)
    pcm = np.round(np.clip(x48, -1, 1 - 1 / 32768) * 32768).astype("<i2")
    n16 = ctx.audio_load(pcm.tobytes(), 1, 16, 1, 48000, 16000)
    assert n16 == 28_800_000
    n, win, hop = pl.window_geometry(n16, 1.0, 0.5)
    assert (n, win, hop) == (3599, WIN, HOP)
    ctx.set_micro_batch(0)
    summ, p1, preds, aligned, res2 = pl.classify_recording(None, c["m1"], c["fx1"], c["m2"], c["fx2"])
    assert summ["num_windows"] == 3599 and np.all(np.isfinite(p1))
    idx = np.array([g for g, _ in res2], np.int64)
    assert len(idx) > 100 and np.array_equal(idx, np.where((p1.argmax(1) == 1) & (p1[:, 1] >= 0.5))[0])
    assert aligned.shape == (3599,) and set(np.unique(aligned)) <= {-1, 0, 1} and (aligned >= 0).sum() == len(idx)
    # a window range run on its own (first_start / different micro-batching) gives the same bits
    audio = ctx.audio_get()
    lo, cnt = 1777, 450
    ctx.logmel(audio[lo * HOP: (lo + cnt - 1) * HOP + WIN], (cnt - 1) * HOP + WIN, 0, HOP, WIN, cnt)
    sub = ctx.softmax(c["m1"].forward_from_slot(cnt))
    assert np.array_equal(sub, p1[lo: lo + cnt])
    # oracle: the device decode + resample + log-mel + forward of the first and the last window
    mono = pcm.astype(np.float32) / 32768.0
    for w in (0, 3598):
        seg48 = mono[max(0, w * HOP * 3 - 3000): (w * HOP + WIN) * 3 + 3000]
        seg16 = orc.resample_sinc_hann(seg48, 48000, 16000)
        off = (w * HOP * 3 - max(0, w * HOP * 3 - 3000)) // 3
        ref_w = seg16[off: off + WIN]
        assert np.abs(ref_w - audio[w * HOP: w * HOP + WIN]).max() <= 2e-6
        ref = orc.softmax(orc.ast_forward(orc.extract_features([audio[w * HOP: w * HOP + WIN]], *S1), orc.ASTWeights(c["sd1"])))
        assert np.abs(p1[w] - ref[0]).max() <= 5e-4

"""Parity at the size the reference actually runs, on the weight set that bites.

The reference loop pushes EVERY window of a recording through stage 1 and the gated ones through stage 2
(src/test_long_audio_windows_2stage.py:301-328); a 30-min file is 3 599 windows (BASELINE.json configs[3]).
tests/golden/sens_tail.npz holds what the REAL transformers classes (ASTFeatureExtractor + ASTForAudioClassification,
fp32, CPU) compute for one such recording on the input-sensitive `sens` weight set (seeds 31 / 33): the stage-1 logits of
all 3 599 windows, the reference's gate at thr1 = 0.5, the stage-2 logits of the gated windows — written by
`tests/golden/make_golden.py --sens-tail` in the build container (about an hour of CPU).  The recording itself is
`synth.synth_recording(seed)`: only outputs are stored.

Here the recording goes through the product's cascade entry point `zk_two_stage` from the AUDIO in every tolerance-meeting
compute mode.  Asserted: max-abs logit error <= 1e-3 everywhere; in the DEFAULT mode (zkast.lib.DEFAULT_COMPUTE_MODE) with at
least 20 % of the tolerance to spare (<= 8e-4); the gate list equals the reference's wherever the fp32 margin is clear.
Printed / written to gpurun_out/sens_tail.json: max, p99.9, p99, median per stage and mode."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOP, WIN = 8000, 16000
TOL = 1e-3
MARGIN = 0.20      # what the default mode must leave of the tolerance at this scale


def _load_tail(golden_dir, name, weight_set):
    from zkast import synth
    g = np.load(os.path.join(golden_dir, name))
    n = int(g["n_windows"])
    rec = synth.synth_recording(int(g["rec_seed"]), WIN + (n - 1) * HOP)
    sd = [synth.make_ast_weights(int(g["s1_seed"]), weight_set), synth.make_ast_weights(int(g["s2_seed"]), weight_set)]
    fx = [(float(g["s1_mean"]), float(g["s1_std"])), (float(g["s2_mean"]), float(g["s2_std"]))]
    return dict(g=g, n=n, rec=rec, sd=sd, fx=fx)


@pytest.fixture(scope="module")
def tail(golden_dir):
    return _load_tail(golden_dir, "sens_tail.npz", "sens")


def test_configs3_sized_recording_on_the_trained_like_set(golden_dir, capsys):
    """The same size on the other hard set: `heavy` (heavy-tailed matrices, LayerNorm gain outliers, massive-activation channels —
    what a fine-tuned ViT looks like to a low-precision GEMM), another recording (seed 19), real transformers fp32
    (tests/golden/heavy_tail.npz, `make_golden.py --heavy-tail`).  Its stage-1 decisions hardly vary with the input, so the
    fixture holds stage 1 for all 3 599 windows and stage 2 for every third, compared forward by forward in the default mode."""
    from zkast import ZkASTConfig, ZkASTForAudioClassification, lib
    path = os.path.join(golden_dir, "heavy_tail.npz")
    if not os.path.exists(path):
        pytest.skip("heavy_tail.npz not generated")
    t = _load_tail(golden_dir, "heavy_tail.npz", "heavy")
    g, n, rec = t["g"], t["n"], t["rec"]
    assert int(g["gated"]) == 0
    ctx = lib.get_context(0)
    ctx.set_micro_batch(0)
    for st in (0, 1):
        ZkASTForAudioClassification(ZkASTConfig(num_labels=2), t["sd"][st], stage=st, compute_mode=lib.DEFAULT_COMPUTE_MODE,
                                    fx_mean=t["fx"][st][0], fx_std=t["fx"][st][1])
    assert ctx.audio_load(rec.tobytes(), 3, 32, 1, 16000, 16000) == rec.size
    ctx.logmel(None, rec.size, 0, HOP, WIN, n)
    s1 = np.empty((n, 2), np.float32)
    ctx.ast_forward(0, None, None, n, s1)
    idx = np.ascontiguousarray(g["swallow_idx"], np.int32)
    s2 = np.empty((len(idx), 2), np.float32)
    ctx.ast_forward(1, None, idx, len(idx), s2)
    e1, e2 = np.abs(s1 - g["s1_logits"]).max(axis=1), np.abs(s2 - g["s2_logits"]).max(axis=1)
    rep = {"set": "heavy", "mode": lib.DEFAULT_COMPUTE_MODE, "windows": n, "stage2_windows": int(len(idx))}
    for name, e in (("stage1", e1), ("stage2", e2)):
        rep[name] = dict(max=float(e.max()), p999=float(np.percentile(e, 99.9)), median=float(np.median(e)), rms=float(np.sqrt((e ** 2).mean())))
    with capsys.disabled():
        print("\n[heavy tail] " + json.dumps(rep))
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        json.dump(rep, open(os.path.join(out_dir, "heavy_tail.json"), "w"), indent=1)
    assert max(rep["stage1"]["max"], rep["stage2"]["max"]) <= (1.0 - MARGIN) * TOL, rep


@pytest.mark.parametrize("mode", ["default", "f16c8", "f16x3"])
def test_configs3_sized_recording_against_real_transformers(tail, mode, capsys):
    from zkast import ZkASTConfig, ZkASTForAudioClassification, lib
    is_default = mode == "default"
    mode = lib.DEFAULT_COMPUTE_MODE if is_default else mode
    if not is_default and mode == lib.DEFAULT_COMPUTE_MODE:
        pytest.skip("covered by the default-mode case")
    t, g, n = tail, tail["g"], tail["n"]
    ctx = lib.get_context(0)
    ctx.set_micro_batch(0)
    for st in (0, 1):
        ZkASTForAudioClassification(ZkASTConfig(num_labels=2), t["sd"][st], stage=st, compute_mode=mode, fx_mean=t["fx"][st][0], fx_std=t["fx"][st][1])
    rec = t["rec"]
    assert ctx.audio_load(rec.tobytes(), 3, 32, 1, 16000, 16000) == rec.size      # load_audio on the device
    s1, idx, s2 = ctx.two_stage(None, rec.size, 0, HOP, WIN, n, float(g["thr1"]))
    ref1, ref_idx, ref2 = g["s1_logits"], g["swallow_idx"], g["s2_logits"]
    assert s1.shape == ref1.shape
    e1 = np.abs(s1 - ref1).max(axis=1)
    # the gate: identical wherever the reference's decision is not a coin flip (p_swallow within 2e-3 of the argmax / thr1 edge)
    m1 = ref1[:, 1] - ref1[:, 0]
    clear = np.abs(m1) > 4e-3
    want = np.zeros(n, bool); want[ref_idx] = True
    got = np.zeros(n, bool); got[idx] = True
    assert np.array_equal(got[clear], want[clear])
    # stage 2 on the windows both gates agree on (all of them unless a window sits on the edge)
    both = np.intersect1d(idx, ref_idx)
    assert len(both) >= len(ref_idx) - int((~clear).sum())
    e2 = np.abs(s2[np.searchsorted(idx, both)] - ref2[np.searchsorted(ref_idx, both)]).max(axis=1)
    rep = {"mode": mode, "windows": n, "gated": int(len(ref_idx))}
    for name, e in (("stage1", e1), ("stage2", e2)):
        rep[name] = dict(max=float(e.max()), p999=float(np.percentile(e, 99.9)), p99=float(np.percentile(e, 99)), median=float(np.median(e)),
                         rms=float(np.sqrt((e ** 2).mean())), worst_window=int(e.argmax()))
    rep["margin_left"] = 1.0 - max(rep["stage1"]["max"], rep["stage2"]["max"]) / TOL
    with capsys.disabled():
        print("\n[sens tail] " + json.dumps(rep))
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        path = os.path.join(out_dir, "sens_tail.json")
        old = json.load(open(path)) if os.path.exists(path) else {}
        old[mode + (" (default)" if is_default else "")] = rep
        json.dump(old, open(path, "w"), indent=1)
    worst = max(rep["stage1"]["max"], rep["stage2"]["max"])
    if mode == "f16c8" and not is_default:
        # measured, not relied on: every layer in f16c8 sits AT the tolerance at this scale (9.5e-4 ... 1.06e-3 on this recording,
        # depending on the build) — the reason it is no longer the default.  The bound only catches a regression.
        assert worst <= 1.25 * TOL, rep
        return
    assert worst <= TOL, rep
    if is_default:
        assert rep["margin_left"] >= MARGIN, rep

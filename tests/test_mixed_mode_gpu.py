"""ZK_F16MIX: one compute mode per encoder layer (zk_model_set_layer_modes).  Layers exchange only the fp32 residual
stream, so a per-layer choice must reproduce the uniform modes bit for bit when every layer gets the same one, and sit
inside the tolerance for any mixture.  The reference has one dtype per model (src/test_long_audio_windows_2stage.py:96)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import ast_oracle as orc  # noqa: E402  (checker only)

S1 = (-1.1509622, 3.5340312)


def _load(wset, seed, mode):
    from zkast import ZkASTConfig, ZkASTForAudioClassification, synth
    sd = synth.make_ast_weights(seed, wset)
    return ZkASTForAudioClassification(ZkASTConfig(num_labels=2), sd, stage=0, compute_mode=mode, fx_mean=S1[0], fx_std=S1[1])


def _slot_logits(ctx, n):
    lg = np.empty((n, 2), np.float32)
    ctx.ast_forward(0, None, None, n, lg)
    return lg


def test_uniform_layer_modes_equal_the_uniform_modes_bit_for_bit():
    from zkast import lib, synth
    ctx = lib.get_context(0)
    rec = synth.synth_recording(5, 16000 + 8 * 8000)
    ctx.logmel(rec, rec.size, 0, 8000, 16000, 9)
    m = _load("sens", 31, "f16c8")
    ref = {}
    for mode in ("f16c8", "f16x3", "f16"):
        m.set_compute_mode(mode)
        ref[mode] = _slot_logits(ctx, 9)
    for mode in ("f16c8", "f16x3"):
        m.set_layer_modes([mode] * 12)
        assert m.compute_mode == "f16mix"
        assert np.array_equal(_slot_logits(ctx, 9), ref[mode]), mode
    # single-pass layers are allowed in a mixture; the patch embedding of a mixture stays the 3-pass one (not ZK_F16's)
    m.set_layer_modes(["f16"] * 12)
    assert 0 < np.abs(_slot_logits(ctx, 9) - ref["f16"]).max() < 2e-2
    assert not np.array_equal(ref["f16c8"], ref["f16x3"])      # (the comparison above can tell the modes apart)
    # the named mode = its documented default assignment; a full (B,1024,128) input takes the same arithmetic
    m.set_compute_mode("f16mix")
    a = _slot_logits(ctx, 9)
    m.set_layer_modes(lib.mix_layer_modes())
    assert np.array_equal(_slot_logits(ctx, 9), a)
    feats = orc.extract_features(orc.window_audio(rec), *S1)
    ctx.set_layer0_attention(0)      # (a caller-provided input takes no layer-0 shortcut; the constant-row attention sums in another order)
    try:
        assert np.abs(m(feats).logits - _slot_logits(ctx, 9)).max() <= 2e-5
    finally:
        ctx.set_layer0_attention(1)


def test_mixed_layers_against_transformers_golden(golden_dir):
    """layer-0 reuse, last-layer pruning and the debug tap inside a mixture; logits against the real transformers run"""
    from zkast import lib, synth
    ctx = lib.get_context(0)
    g = np.load(os.path.join(golden_dir, "model_sens.npz"))
    wins = synth.golden_windows()
    rec = np.concatenate(list(wins))
    ctx.logmel(rec, rec.size, 0, 16000, 16000, 6)
    m = _load("sens", 31, "f16c8")
    X, C8, F = "f16x3", "f16c8", "f16"
    cases = [[X] * 4 + [C8] * 8, [C8] * 11 + [X], [X, C8] * 6, [C8, X] * 6,
             # per kernel group (qkv, att, o, mlp): every combination the plane formats allow, in the first, a middle and the last layer
             [(X, X, C8, C8)] + [C8] * 10 + [(C8, X, X, C8)], [(C8, X, C8, X)] + [C8] * 5 + [(X, X, X, C8)] + [C8] * 4 + [(C8, C8, X, X)],
             [(C8, C8, C8, X), (X, X, C8, X)] * 6, [(C8, F, X, C8)] + [C8] * 10 + [(X, F, C8, X)]]
    for modes in cases:
        m.set_layer_modes(modes)
        lg = _slot_logits(ctx, 6)
        err = np.abs(lg - g["sens_logits"]).max()
        tag = " ".join(v[4] if isinstance(v, str) else "".join(k[4] if len(k) > 3 else "1" for k in v) for v in modes)
        print(f"[mixed {tag}] max-abs logit err vs transformers fp32: {err:.3e}")
        if F in str(modes):      # (a single-pass QK^T is outside the tolerance by design: it only has to run)
            assert err <= 2e-2
            continue
        assert err <= 1e-3
        ctx.set_layer0_attention(0)      # (the constant-row attention state sums in another order: tests/test_model_gpu.py)
        lg = _slot_logits(ctx, 6)
        for flag in (0, 1):      # the exact shortcuts stay exact inside a mixture
            ctx.set_layer0_reuse(flag)
            ctx.set_prune_last_layer(flag)
            assert np.array_equal(_slot_logits(ctx, 6), lg), (modes, flag)
        ctx.set_layer0_reuse(1)
        ctx.set_prune_last_layer(1)
        ctx.set_layer0_attention(1)
    # residual-stream checkpoint after a c8 layer that follows x3 layers
    m.set_layer_modes(["f16x3"] * 5 + ["f16c8"] * 7)
    ctx.debug_tap(5)
    _slot_logits(ctx, 6)
    h = ctx.debug_get_tap(6)
    ctx.debug_tap(-2)
    ref_tok = g["sens_layer5_tok"]
    assert np.abs(h[:, g["tokens"]] - ref_tok).max() <= 2e-4 * np.abs(ref_tok).max()


def test_layer_modes_argument_errors():
    from zkast import lib
    m = _load("init", 12, "f16c8")
    with pytest.raises(lib.ZkError, match="12 modes"):
        m.set_layer_modes(["f16c8"] * 11)
    with pytest.raises(lib.ZkError, match="layer 3"):      # c8 QK^T needs the c8 QKV epilogue's k plane
        m.set_layer_modes(["f16c8"] * 3 + [("f16x3", "f16c8", "f16c8", "f16c8")] + ["f16c8"] * 8)
    with pytest.raises(lib.ZkError, match="layer 0"):      # a split QK^T needs a lo plane
        m.set_layer_modes([("f16", "f16x3", "f16c8", "f16c8")] + ["f16c8"] * 11)
    with pytest.raises(ValueError):
        m.set_layer_modes([("f16c8", "f16c8")] * 12)
    with pytest.raises(KeyError):
        m.set_layer_modes(["bf16"] * 12)
    ctx = lib.get_context(0)
    import ctypes as C
    bad = (C.c_int32 * 12)(*([2] * 11 + [4]))      # ZK_F16MIX is not a per-layer mode
    assert ctx.lib.zk_model_set_layer_modes(ctx.h, 0, bad, 12) == -1
    assert b"layer 11" in ctx.lib.zk_last_error(ctx.h)

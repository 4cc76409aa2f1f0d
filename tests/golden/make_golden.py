#!/usr/bin/env python
"""Generate the golden fixtures in this directory.  RUNS ONLY IN THE BUILD CONTAINER (needs /root/reference and
the `transformers` package); the fixtures it writes are data (inputs + expected outputs) and are committed, this
script never runs on the GPU box.

Sources of truth:
  * `transformers.ASTFeatureExtractor` / `ASTForAudioClassification` constructed locally (no hub access) — the
    third-party code the reference calls (src/test_long_audio_windows_2stage.py:40,89-94,108-110);
  * the reference module's own pure functions `window_audio`, `forward_probs`, `summarize_stage_outputs`
    (src/test_long_audio_windows_2stage.py:62-75,104-113,148-195 and ..._cache.py:243-297), imported from
    /root/reference with an empty `torchaudio` stub (torchaudio is not installed; only load_audio/discover use it).

Weights come from zkast.synth (splitmix64), so only outputs are stored.

    python tests/golden/make_golden.py
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "zenker-audio-detection_amd"))

from transformers import ASTConfig, ASTFeatureExtractor, ASTForAudioClassification  # noqa: E402

from zkast import synth  # noqa: E402

REF = "/root/reference/src"
S1_MEAN, S1_STD = -1.1509622, 3.5340312      # src/train_ast_stage1_cross_validation.py:104-105
S2_MEAN, S2_STD = -6.5, 2.75                 # a per-fold style override (train_ast_stage2…:512-514)
TOKENS = [0, 1, 2, 102, 103, 1213]


def load_ref(name):
    if "torchaudio" not in sys.modules:
        sys.modules["torchaudio"] = types.ModuleType("torchaudio")
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def hf_model(seed, weight_set):
    m = ASTForAudioClassification(ASTConfig(num_labels=2)).eval()
    sd = {k: torch.from_numpy(v) for k, v in synth.make_ast_weights(seed, weight_set).items()}
    missing, unexpected = m.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    return m


def make_heavy(nrm):
    """F3b: the "heavy" weight set (heavy-tailed matrices, LayerNorm gain outliers, massive-activation channels) through
    the real transformers fp32 model: logits, residual-stream checkpoints, and the largest |input| every GEMM kind of
    the stack sees (what an fp8 value byte with a fixed scale would have to cover)."""
    feats4 = nrm[[0, 1, 2, 4]]
    f = {"input_windows": np.array([0, 1, 2, 4]), "tokens": np.array(TOKENS)}
    m = hf_model(13, "heavy")
    amax = {"qkv_in": [], "o_in": [], "fc1_in": [], "fc2_in": []}
    hooks = []
    for layer in m.audio_spectrogram_transformer.layers:
        att = layer.attention
        hooks.append(att.q_proj.register_forward_pre_hook(lambda _m, a: amax["qkv_in"].append(float(a[0].abs().max()))))
        hooks.append(att.o_proj.register_forward_pre_hook(lambda _m, a: amax["o_in"].append(float(a[0].abs().max()))))
        hooks.append(layer.mlp.fc1.register_forward_pre_hook(lambda _m, a: amax["fc1_in"].append(float(a[0].abs().max()))))
        hooks.append(layer.mlp.fc2.register_forward_pre_hook(lambda _m, a: amax["fc2_in"].append(float(a[0].abs().max()))))
    out = m(torch.from_numpy(feats4), output_hidden_states=True)
    for h in hooks:
        h.remove()
    hs = out.hidden_states
    seq = m.audio_spectrogram_transformer.layernorm(hs[-1])
    f["heavy_logits"] = out.logits.numpy()
    for nm, t in [("emb", hs[0]), ("layer0", hs[1]), ("layer5", hs[6]), ("layer11", hs[12]), ("final_ln", seq)]:
        f[f"heavy_{nm}_tok"] = t[:, TOKENS].numpy()
        f[f"heavy_{nm}_norm"] = t.norm(dim=-1).numpy()
    f["heavy_resid_absmax"] = np.array([float(h.abs().max()) for h in hs])
    for k, v in amax.items():
        f[f"heavy_{k}_absmax"] = np.array(v)
    print("heavy logits", out.logits.numpy().tolist())
    print("heavy |x| max per GEMM input kind:", {k: round(max(v), 2) for k, v in amax.items()},
          "residual stream:", round(float(f["heavy_resid_absmax"].max()), 1))
    np.savez_compressed(os.path.join(HERE, "model_heavy.npz"), **f)


def make_sens(nrm):
    """F3c: the INPUT-SENSITIVE weight set ("sens", seed 31: patch filters dominate the embedding, content-peaked
    attention, 2x head gain) through the real transformers fp32 model on all six golden windows: logits that span > 6
    between windows and whose argmax flips, plus the usual residual-stream checkpoints."""
    f = {"input_windows": np.arange(6), "tokens": np.array(TOKENS)}
    m = hf_model(31, "sens")
    out = m(torch.from_numpy(nrm[:6]), output_hidden_states=True)
    hs = out.hidden_states
    seq = m.audio_spectrogram_transformer.layernorm(hs[-1])
    lg = out.logits.numpy()
    f["sens_logits"] = lg
    for nm, t in [("emb", hs[0]), ("layer0", hs[1]), ("layer5", hs[6]), ("layer11", hs[12]), ("final_ln", seq)]:
        f[f"sens_{nm}_tok"] = t[:, TOKENS].numpy()
        f[f"sens_{nm}_norm"] = t.norm(dim=-1).numpy()
    margin = lg[:, 1] - lg[:, 0]
    print("sens logits", lg.tolist(), "margins", margin.tolist())
    assert np.ptp(lg, axis=0).max() > 3.0 and margin.min() < -1.0 and margin.max() > 1.0, "set is not input-sensitive"
    np.savez_compressed(os.path.join(HERE, "model_sens.npz"), **f)


SENS_TAIL_N, SENS_TAIL_REC_SEED = 3599, 17


def _fx_chunk(args):
    """worker of make_sens_tail: the real extractor on one chunk of windows (one process per chunk: the extractor is a
    serial Python loop per window)."""
    mean, std, wins = args
    torch.set_num_threads(1)
    fx = ASTFeatureExtractor(mean=mean, std=std)
    return fx(list(wins), sampling_rate=16000, return_tensors="np")["input_values"]


def make_sens_tail(n=SENS_TAIL_N, rec_seed=SENS_TAIL_REC_SEED, procs=6, threads=6, weight_set="sens", seeds=(31, 33), out_name="sens_tail.npz",
                   gate=True):
    """F7: ONE configs[3]-sized recording (30 min at 16 kHz -> 3 599 windows of 1 s / 0.5 s hop, the count the reference
    loop pushes through both stages per file, src/test_long_audio_windows_2stage.py:301-328) through the REAL
    `ASTFeatureExtractor` + `ASTForAudioClassification` on the input-sensitive `sens` weight set (seeds 31 / 33, the two
    stages of tests/test_sens_batch_gpu.py): stage-1 logits of every window, the reference's gate at thr1 = 0.5 (:312-320),
    stage-2 logits of the gated windows.  Only outputs are stored; the recording is `synth.synth_recording(rec_seed, ...)`
    windowed by the reference's own `window_audio`.  About an hour of CPU in the build container."""
    import multiprocessing as mp
    import time
    ref = load_ref("test_long_audio_windows_2stage")
    ref.DEVICE = torch.device("cpu")
    rec = synth.synth_recording(rec_seed, 16000 + (n - 1) * 8000)
    wins = ref.window_audio(rec, 1.0, 0.5)
    assert len(wins) == n and all(len(w) == 16000 for w in wins)
    torch.set_num_threads(threads)
    pool = mp.get_context("spawn").Pool(procs)      # spawn: the extractor pads with torch operators, which a fork of a threaded parent deadlocks
    part = os.path.join("/tmp", f"{weight_set}_tail_{rec_seed}_{n}.partial.npz")
    done = dict(np.load(part)) if os.path.exists(part) else {}

    def run_stage(tag, seed, mean, std, sel):
        key = f"{tag}_logits"
        if key in done and len(done[key]) == len(sel):
            return done[key]
        m = hf_model(seed, weight_set)
        out = np.zeros((len(sel), 2), np.float32)
        t0 = time.time()
        bs = 64 * procs
        if True:
            for lo in range(0, len(sel), bs):
                idx = sel[lo:lo + bs]
                chunks = [(mean, std, [wins[i] for i in idx[c::procs]]) for c in range(procs) if len(idx[c::procs])]
                feats = pool.map(_fx_chunk, chunks)
                for c, f in enumerate(feats):
                    rows = np.arange(lo, lo + len(idx))[c::procs]
                    for b in range(0, len(f), 16):
                        out[rows[b:b + 16]] = m(torch.from_numpy(f[b:b + 16])).logits.numpy()
                print(f"{tag}: {lo + len(idx)} / {len(sel)} windows, {time.time() - t0:.0f} s", flush=True)
        done[key] = out
        np.savez(part, **done)
        return out

    s1 = run_stage("s1", seeds[0], S1_MEAN, S1_STD, np.arange(n))
    p1 = torch.softmax(torch.from_numpy(s1), dim=1).numpy()            # forward_probs :111
    pred = np.where((p1.argmax(axis=1) == 1) & (p1[:, 1] >= 0.5), 1, 0)  # the gate, :312-320
    # gate = False (weight sets whose stage-1 decisions do not vary with the input: the gate would pass all windows or none):
    # stage 2 on every third window instead, compared forward by forward (no cascade call)
    idx = np.where(pred == 1)[0] if gate else np.arange(0, n, 3)
    s2 = run_stage("s2", seeds[1], S2_MEAN, S2_STD, idx)
    pool.close()
    margin = s1[:, 1] - s1[:, 0]
    print(f"{weight_set} tail: {n} windows, {len(idx)} through stage 2 ({'the gate' if gate else 'every third'}), stage-1 margin span "
          f"{margin.min():.2f} .. {margin.max():.2f}")
    np.savez_compressed(os.path.join(HERE, out_name), s1_logits=s1, swallow_idx=idx.astype(np.int32), s2_logits=s2,
                        rec_seed=rec_seed, n_windows=n, s1_seed=seeds[0], s2_seed=seeds[1], thr1=0.5, gated=int(gate), weight_set=weight_set,
                        s1_mean=S1_MEAN, s1_std=S1_STD, s2_mean=S2_MEAN, s2_std=S2_STD)


def main():
    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    if "--heavy-tail" in sys.argv:      # adds heavy_tail.npz: the same recording size on the trained-like `heavy` set (another recording)
        make_sens_tail(n=int(os.environ.get("SENS_TAIL_N", SENS_TAIL_N)), rec_seed=19, weight_set="heavy", seeds=(13, 14),
                       out_name="heavy_tail.npz", gate=False)
        return
    if "--sens-tail" in sys.argv:       # adds sens_tail.npz (about an hour) without touching the other fixtures
        make_sens_tail(n=int(os.environ.get("SENS_TAIL_N", SENS_TAIL_N)))     # other N: plumbing check, writes the same file name
        return
    if "--sens-only" in sys.argv:       # adds model_sens.npz without touching the other fixtures
        fx = ASTFeatureExtractor(mean=S1_MEAN, std=S1_STD)
        nrm = fx(list(synth.golden_windows()), sampling_rate=16000, return_tensors="np")["input_values"]
        make_sens(nrm)
        return
    if "--heavy-only" in sys.argv:      # adds model_heavy.npz without touching the other fixtures
        fx = ASTFeatureExtractor(mean=S1_MEAN, std=S1_STD)
        nrm = fx(list(synth.golden_windows()), sampling_rate=16000, return_tensors="np")["input_values"]
        make_heavy(nrm)
        return
    ref = load_ref("test_long_audio_windows_2stage")
    refc = load_ref("test_long_audio_windows_2stage_cache")

    # ---------------- F1: window indexing ----------------
    f1 = {}
    for T in [5000, 16000, 16001, 23999, 24000, 40000, 28_800_000]:
        audio = np.zeros(T, dtype=np.float32)
        audio[: min(T, 100)] = 1.0
        wins = ref.window_audio(audio, 1.0, 0.5)
        f1[str(T)] = {"n": len(wins), "lens": sorted({int(len(w)) for w in wins}),
                      "first_sum": float(wins[0].sum()), "last_sum": float(wins[-1].sum())}
    f1["hop_gt_win"] = {"n": len(ref.window_audio(np.zeros(80000, np.float32), 1.0, 1.5))}
    f1["win_0p25_hop_0p1"] = {"n": len(ref.window_audio(np.zeros(16000, np.float32), 0.25, 0.1))}
    json.dump(f1, open(os.path.join(HERE, "windows.json"), "w"), indent=1)

    # ---------------- F2: log-mel ----------------
    wins = synth.golden_windows()
    fx = ASTFeatureExtractor(mean=S1_MEAN, std=S1_STD)
    fx_raw = ASTFeatureExtractor(mean=S1_MEAN, std=S1_STD, do_normalize=False)
    raw = fx_raw(list(wins), sampling_rate=16000, return_tensors="np")["input_values"]
    nrm = fx(list(wins), sampling_rate=16000, return_tensors="np")["input_values"]
    assert raw.shape == (6, 1024, 128) and raw.dtype == np.float32
    assert np.all(raw[:, 98:] == 0.0)
    # short input (fewer than 98 frames) and long input (2 s -> 198 frames): pad/frames logic
    short = fx_raw(wins[0][:4000], sampling_rate=16000, return_tensors="np")["input_values"][0]
    long2 = fx_raw(np.concatenate([wins[0], wins[2]]), sampling_rate=16000, return_tensors="np")["input_values"][0]
    np.savez_compressed(
        os.path.join(HERE, "fbank.npz"),
        raw_rows=raw[:, :98].copy(), norm_rows=nrm[:, :98].copy(), norm_pad_value=nrm[0, 500, 0],
        mel_filters=fx.mel_filters.astype(np.float64), window=fx.window.astype(np.float64),
        short_rows=short[:30].copy(), short_n=int((np.abs(short).sum(1) != 0).sum()),
        long_rows=long2[:198:9].copy(), long_n=int((np.abs(long2).sum(1) != 0).sum()),
        mean=S1_MEAN, std=S1_STD,
    )
    try:
        fx(list(wins[:1]), sampling_rate=8000)
        raise SystemExit("expected ValueError")
    except ValueError as e:
        sr_err = str(e)[:60]

    # ---------------- F3: model ----------------
    feats4 = nrm[[0, 1, 2, 4]]
    f3 = {"input_windows": np.array([0, 1, 2, 4])}
    for tag, seed, wset in [("wide", 11, "wide"), ("init", 12, "init")]:
        m = hf_model(seed, wset)
        out = m(torch.from_numpy(feats4), output_hidden_states=True)
        hs = out.hidden_states           # embeddings output + each layer output (13)
        assert len(hs) == 13 and hs[0].shape == (4, 1214, 768)
        base = m.audio_spectrogram_transformer
        seq = base.layernorm(hs[-1])
        pooled = (seq[:, 0] + seq[:, 1]) / 2
        f3[f"{tag}_logits"] = out.logits.numpy()
        f3[f"{tag}_pooled"] = pooled.numpy()
        for nm, t in [("emb", hs[0]), ("layer0", hs[1]), ("layer5", hs[6]), ("layer11", hs[12]), ("final_ln", seq)]:
            f3[f"{tag}_{nm}_tok"] = t[:, TOKENS].numpy()
            f3[f"{tag}_{nm}_norm"] = t.norm(dim=-1).numpy()
        print(tag, "logits", out.logits.numpy().tolist())
    f3["tokens"] = np.array(TOKENS)
    np.savez_compressed(os.path.join(HERE, "model.npz"), **f3)
    make_heavy(nrm)
    make_sens(nrm)

    # ---------------- F4: cascade ----------------
    w16 = synth.synth_windows(seed=3, n_windows=16)
    m1 = hf_model(21, "wide")
    m2 = hf_model(22, "wide")
    fx1 = ASTFeatureExtractor(mean=S1_MEAN, std=S1_STD)
    fx2 = ASTFeatureExtractor(mean=S2_MEAN, std=S2_STD)
    ref.DEVICE = torch.device("cpu")
    # random weights give near-constant decisions; centre each head on the median logit gap so that the gate
    # splits the 16 windows.  The shift is stored and re-applied to the synth weights by the tests.
    shifts = []
    for m, fxs in ((m1, fx1), (m2, fx2)):
        lg = m(fxs(list(w16), sampling_rate=16000, return_tensors="pt")["input_values"]).logits.numpy()
        sh = float(np.median(lg[:, 0] - lg[:, 1]))
        m.classifier.dense.bias.data[1] += sh
        shifts.append(sh)
    p1 = ref.forward_probs(m1, fx1, list(w16), 5)       # ragged batches 5,5,5,1
    p2 = ref.forward_probs(m2, fx2, list(w16), 16)
    empty = ref.forward_probs(m1, fx1, [], 5)
    assert p1.shape == (16, 2) and p1.dtype == np.float32 and empty.shape == (0,)
    np.savez_compressed(os.path.join(HERE, "cascade.npz"), s1_probs=p1, s2_probs_all=p2,
                        s1_mean=S1_MEAN, s1_std=S1_STD, s2_mean=S2_MEAN, s2_std=S2_STD,
                        s1_bias_shift=shifts[0], s2_bias_shift=shifts[1], s1_seed=21, s2_seed=22, audio_seed=3)
    print("s1 probs[:,1]", p1[:, 1].round(4).tolist())

    # summaries on hand-made tables (incl. argmax-vs-threshold quirk, empty, all-swallow)
    rng = np.random.default_rng(5)
    cases = {}

    def run_case(name, s1, thr1, thr2, s2_all, min_prob=None, use_argmax=False):
        s1 = np.asarray(s1, dtype=np.float32)
        s2_all = np.asarray(s2_all, dtype=np.float32)
        pred = s1.argmax(axis=1)
        pred = np.where((pred == 1) & (s1[:, 1] >= thr1), 1, 0)
        idx = np.where(pred == 1)[0]
        if min_prob is not None:
            idx = idx[s1[idx, 1] >= min_prob]
        res = [(int(i), s2_all[i]) for i in idx]
        if use_argmax or min_prob is not None:
            summ = refc.summarize_stage_outputs(s1, res, ["Idle", "Swallow"], ["Healthy", "Zenker"], thr2, use_argmax)
        else:
            summ = ref.summarize_stage_outputs(s1, res, ["Idle", "Swallow"], ["Healthy", "Zenker"], thr2)
        cases[name] = {"s1": s1.tolist(), "s2_all": s2_all.tolist(), "thr1": thr1, "thr2": thr2,
                       "min_prob": min_prob, "use_argmax": use_argmax, "swallow_idx": idx.tolist(), "summary": summ}

    def rand_probs(n):
        a = rng.uniform(0.02, 0.98, n)
        return np.stack([1 - a, a], 1)

    s1 = rand_probs(12)
    s2 = rand_probs(12)
    run_case("thr_0p5", s1, 0.5, 0.5, s2)
    run_case("thr_0p55_quirk", [[0.48, 0.52], [0.3, 0.7], [0.9, 0.1], [0.46, 0.54]], 0.55, 0.5,
             [[0.1, 0.9], [0.2, 0.8], [0.5, 0.5], [0.6, 0.4]])
    run_case("thr_0p9", s1, 0.9, 0.35, s2)
    run_case("no_swallow", [[0.9, 0.1], [0.8, 0.2], [0.5, 0.5]], 0.5, 0.5, [[0.5, 0.5]] * 3)   # tie -> idle
    run_case("all_swallow", [[0.1, 0.9], [0.2, 0.8]], 0.5, 0.5, [[0.7, 0.3], [0.5, 0.5]])     # p==thr2 -> zenker
    run_case("argmax_none_evaluated", [[0.4, 0.6], [0.45, 0.55]], 0.7, 0.5, [[0.5, 0.5]] * 2)
    run_case("cache_argmax", s1, 0.5, 0.8, s2, use_argmax=True)
    run_case("cache_min_prob", s1, 0.5, 0.5, s2, min_prob=0.75)
    # ---------------- F6: patient-level aggregation (utils/aggregate_2stage_results.py) + batch-driver helpers -------
    import contextlib, io, tempfile
    spec = importlib.util.spec_from_file_location("agg_ref", "/root/reference/utils/aggregate_2stage_results.py")
    agg_ref = importlib.util.module_from_spec(spec); sys.modules["agg_ref"] = agg_ref; spec.loader.exec_module(agg_ref)
    spec = importlib.util.spec_from_file_location("batch_ref", os.path.join(REF, "run_batch_simple_2stage.py"))
    batch_ref = importlib.util.module_from_spec(spec); sys.modules["batch_ref"] = batch_ref; spec.loader.exec_module(batch_ref)
    patients = [("101", "Healthy", 0.10, 40), ("102", "Healthy", 0.60, 30), ("103", "Zenker", 0.50, 25),
                ("104", "Zenker", 0.20, 50), ("105", "Zenker", None, 0), ("106", "Other", 0.9, 10),
                ("107", "Zenker", 0.95, 12)]
    agg_inputs, agg_expected = {}, {}
    with tempfile.TemporaryDirectory() as td:
        for pid, cls, ratio, sw in patients:
            doc = {"aggregate": {"files_used": [f"/data/Long/{cls}/{pid}/a.wav", f"/data/Long/{cls}/{pid}/b.wav"],
                                 "total_windows": 100, "total_swallow_windows": sw,
                                 "total_zenker_windows": 0 if ratio is None else int(round(ratio * sw)),
                                 "total_healthy_windows": 0, "overall_zenker_ratio_over_swallow": ratio}}
            agg_inputs[pid] = doc
            json.dump(doc, open(os.path.join(td, f"{pid}_2stage.json"), "w"))
        json.dump({"x": 1}, open(os.path.join(td, "batch_fold1_2stage.json"), "w"))
        for thr in (0.5, 0.3):
            ns = types.SimpleNamespace(outputs_dir=td, threshold=thr, csv=None, json=None, verbose=False, store_output=False)
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                agg_ref.aggregate(ns)
            summ = json.loads(buf.getvalue())
            summ.pop("outputs_dir")
            agg_expected[str(thr)] = summ
        ids_txt = "Healthy/224\n\nZenker/006\n  Healthy/Sub/31  \n"
        open(os.path.join(td, "ids.txt"), "w").write(ids_txt)
        ids_expected = batch_ref.read_ids(os.path.join(td, "ids.txt"))
    thr_cfg = {"folds": {"1": {"stage1": {"threshold": 0.61}, "stage2": {"threshold": 0.42}}, "2": {"stage2": {"threshold": 0.7}}},
               "thresholds": {"stage1": {"threshold": 0.55}}}
    thr_expected = {}
    for fold in (1, 2, 3):
        ns = types.SimpleNamespace(long_audio_root="/d", pattern="*.wav", fold=fold, window_sec=1.0, hop_sec=0.5,
                                   stage1_model_root=None, stage2_model_root=None, stage1_forward_min_prob=None,
                                   stage2_argmax=False, output_dir=None, plot=False, extra=None)
        cmd = batch_ref.build_cmd(ns, "006", thr_cfg)
        thr_expected[str(fold)] = {k[2:].replace("-", "_"): float(cmd[cmd.index(k) + 1])
                                   for k in ("--stage1-threshold", "--stage2-threshold") if k in cmd}
    json.dump({"agg_inputs": agg_inputs, "agg_expected": agg_expected, "ids_text": ids_txt, "ids_expected": ids_expected,
               "threshold_config": thr_cfg, "thresholds_expected": thr_expected},
              open(os.path.join(HERE, "batch_aggregate.json"), "w"), indent=1)

    fx_dict = ASTFeatureExtractor(mean=S1_MEAN, std=S1_STD).to_dict()
    fx_fp = refc.get_fx_fingerprint(ASTFeatureExtractor(mean=S1_MEAN, std=S1_STD))
    meta = {"sampling_rate_error_prefix": sr_err, "cases": cases, "fx_to_dict": fx_dict, "fx_fingerprint": fx_fp,
            "transformers": __import__("transformers").__version__, "torch": torch.__version__,
            "numpy": np.__version__}
    json.dump(meta, open(os.path.join(HERE, "cascade_cases.json"), "w"), indent=1, default=lambda o: float(o))
    print("done")


if __name__ == "__main__":
    main()

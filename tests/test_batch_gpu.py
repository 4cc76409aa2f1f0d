"""BASELINE config 5 in miniature on a real MI355X: synthetic patients (two 48 kHz PCM16 WAV files each, so the
device resampler is on the path), model directories on disk in the layout trainer.save_model writes
(config.json + preprocessor_config.json + model.safetensors), the in-process batch driver, per-patient JSON in the
reference schema, patient-level aggregation — checked against the CPU oracle run on the same files."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import ast_oracle as orc  # noqa: E402  (checker only)


def _save_model_dir(path, sd, mean, std, labels):
    from safetensors.numpy import save_file
    os.makedirs(path, exist_ok=True)
    save_file({k: np.ascontiguousarray(v) for k, v in sd.items()}, os.path.join(path, "model.safetensors"))
    json.dump({"architectures": ["ASTForAudioClassification"], "hidden_size": 768, "num_hidden_layers": 12,
               "num_attention_heads": 12, "intermediate_size": 3072, "patch_size": 16, "frequency_stride": 10,
               "time_stride": 10, "max_length": 1024, "num_mel_bins": 128, "layer_norm_eps": 1e-12,
               "hidden_act": "gelu", "qkv_bias": True, "id2label": {"0": labels[0], "1": labels[1]},
               "label2id": {labels[0]: 0, labels[1]: 1}}, open(os.path.join(path, "config.json"), "w"))
    json.dump({"feature_extractor_type": "ASTFeatureExtractor", "do_normalize": True, "mean": mean, "std": std,
               "max_length": 1024, "num_mel_bins": 128, "sampling_rate": 16000, "padding_value": 0.0,
               "feature_size": 1, "return_attention_mask": False}, open(os.path.join(path, "preprocessor_config.json"), "w"))


def test_batch_driver_end_to_end(tmp_path):
    from zkast import aggregate, batch, pipeline, synth
    cas = np.load(os.path.join(os.path.dirname(__file__), "golden", "cascade.npz"))
    sd1 = synth.make_ast_weights(int(cas["s1_seed"]), "wide")
    sd1["classifier.dense.bias"][1] += np.float32(cas["s1_bias_shift"])
    sd2 = synth.make_ast_weights(int(cas["s2_seed"]), "wide")
    sd2["classifier.dense.bias"][1] += np.float32(cas["s2_bias_shift"])
    S1, S2 = (float(cas["s1_mean"]), float(cas["s1_std"])), (float(cas["s2_mean"]), float(cas["s2_std"]))
    m1, m2 = str(tmp_path / "runs" / "s1"), str(tmp_path / "runs" / "s2")
    _save_model_dir(m1, sd1, *S1, ["Idle", "Swallow"])
    _save_model_dir(m2, sd2, *S2, ["Healthy", "Zenker"])
    root = tmp_path / "Long"
    ids = []
    for pid, cls, seed in [("201", "Healthy", 3), ("202", "Zenker", 40)]:
        d = root / cls / pid
        d.mkdir(parents=True)
        for k in range(2):
            x = synth.synth_recording(seed + k, 48000 * 3 + 1234)      # ~3 s at 48 kHz -> 5 windows after resampling
            pipeline.write_wav_pcm16(str(d / f"rec{k}.wav"), x, 48000)
        ids.append(f"{cls}/{pid}")
    (tmp_path / "ids").mkdir()
    (tmp_path / "ids" / "test_ids_fold1.txt").write_text("\n".join(ids) + "\n")
    json.dump({"folds": {"1": {"stage1": {"threshold": 0.5}, "stage2": {"threshold": 0.45}}}},
              open(tmp_path / "thr.json", "w"))
    out = str(tmp_path / "out")
    st = batch.main(["--fold", "1", "--ids-root", str(tmp_path / "ids"), "--long-audio-root", str(root),
                     "--output-dir", out, "--threshold-config", str(tmp_path / "thr.json"),
                     "--stage1-model-root", m1, "--stage2-model-root", m2])
    assert st == {"201": "ok", "202": "ok"}
    assert batch.main(["--fold", "1", "--ids-root", str(tmp_path / "ids"), "--long-audio-root", str(root),
                       "--output-dir", out, "--stage1-model-root", m1, "--stage2-model-root", m2]) == {"201": "skip", "202": "skip"}
    W1, W2 = orc.ASTWeights(sd1), orc.ASTWeights(sd2)
    for pid, cls in [("201", "Healthy"), ("202", "Zenker")]:
        doc = json.load(open(os.path.join(out, f"{pid}_2stage.json")))
        assert set(doc) == {"config", "per_file", "aggregate"} and doc["config"]["stage1_threshold"] == 0.5
        files = doc["aggregate"]["files_used"]
        assert len(files) == 2 and f"/{cls}/{pid}/" in files[0]
        per_file = {}
        for k, path in enumerate(files):
            wav, sr = pipeline.read_wav(path)
            audio = orc.resample_sinc_hann(wav[0], sr, 16000)
            wins = orc.window_audio(audio)
            p1 = orc.forward_probs(W1, *S1, wins, 8)
            idx = orc.stage1_gate(p1, np.float32(0.5))
            p2 = orc.forward_probs(W2, *S2, [wins[i] for i in idx], 8) if len(idx) else np.zeros((0, 2))
            ref = orc.summarize_stage_outputs(p1, [(int(i), p2[j]) for j, i in enumerate(idx)], 0.45)
            got = doc["per_file"][f"file_{k}"]
            assert got["path"] == path
            for key, v in ref.items():
                if isinstance(v, (int, type(None))):
                    assert got[key] == v, (pid, k, key)
                else:
                    assert np.allclose(got[key], v, atol=5e-4), (pid, k, key)
            per_file[f"file_{k}"] = ref
        ref_agg = pipeline.aggregate_files(per_file, files)
        for key, v in ref_agg.items():
            if isinstance(v, float):
                assert doc["aggregate"][key] == pytest.approx(v, abs=1e-9)
            else:
                assert doc["aggregate"][key] == v
    summ, rows = aggregate.aggregate(out, 0.5)
    assert summ["num_patient_results"] == 2 and {r["gt"] for r in rows} == {"Healthy", "Zenker"}


def test_snippet_level_evaluation_loop(tmp_path):
    """run_inference (analyze_ROC_PR_stage1.py:163-191) and the Trainer.predict contract of
    test_trained_model_stage1_cv.py at batch 8 over snippets of UNEQUAL length and mixed payload kinds (ndarray,
    dict at 16 kHz, dict at 48 kHz -> device resampler, WAV path), checked against the CPU oracle."""
    from zkast import ZkASTConfig, ZkASTFeatureExtractor, ZkASTForAudioClassification, evaluate as ev, pipeline as pl, synth
    sd = synth.make_ast_weights(41, "wide")
    mean, std = -1.1509622, 3.5340312
    mdir = str(tmp_path / "fold0" / "best")
    _save_model_dir(mdir, sd, mean, std, ["Idle", "Swallow"])
    lens = [16000, 12000, 16000, 9000, 16000, 16000, 12000, 4000, 16000, 16000, 7000]
    X, ref_w = [], []
    for i, n in enumerate(lens):
        w = synth.synth_recording(100 + i, n)
        ref_w.append(w)
        if i % 4 == 1:
            X.append({"array": w.tolist(), "sampling_rate": 16000})
        elif i % 4 == 2:
            path = str(tmp_path / f"s{i}.wav")
            pl.write_wav_pcm16(path, w, 16000)
            ref_w[-1] = pl.read_wav(path)[0][0]
            X.append(path)
        else:
            X.append(w)
    scores = ev.run_inference(mdir, X, batch_size=8)
    W = orc.ASTWeights(sd)
    ref = np.concatenate([orc.softmax(orc.ast_forward(orc.extract_features([w], mean, std), W)) for w in ref_w])
    assert scores.shape == (len(lens),) and scores.dtype == np.float32
    assert np.abs(scores - ref[:, 1]).max() <= 1e-3
    # the predict contract: logits in dataset order, argmax + confusion matrix
    fx = ZkASTFeatureExtractor.from_pretrained(mdir)
    model = ZkASTForAudioClassification.from_pretrained(mdir, config=ZkASTConfig.from_pretrained(mdir))
    logits = ev.predict_logits(model, fx, X, batch_size=8)
    y_true = [int(r[1] > r[0]) for r in ref]
    y_pred, cm = ev.evaluate_predictions(logits, y_true, 2)
    assert logits.shape == (len(lens), 2) and y_pred.tolist() == y_true and cm.trace() == len(lens)
    assert ev.run_inference(mdir, [], 8).shape == (0,)
    # a 48 kHz dict payload goes through the device resampler
    w48 = synth.synth_recording(7, 48000)
    s48 = ev.run_inference(mdir, [{"array": w48, "sampling_rate": 48000}], 8)
    w16 = orc.resample_sinc_hann(w48, 48000, 16000)
    r48 = orc.softmax(orc.ast_forward(orc.extract_features([w16], mean, std), W))[0, 1]
    assert abs(float(s48[0]) - float(r48)) <= 1e-3


def test_cli_with_feature_cache_equals_the_uncached_run(tmp_path):
    """The cached variant's options on the command line (--feature-cache-dir / --refresh-cache / --disable-cache,
    src/test_long_audio_windows_2stage_cache.py:361-375): same JSON as the uncached run; the second run is served by the
    compact stores; with only the reference-format `.pt` twins left in the directory (what the reference script writes)
    the run imports those and still agrees."""
    import glob
    from zkast import pipeline, synth
    cas = np.load(os.path.join(os.path.dirname(__file__), "golden", "cascade.npz"))
    sd1 = synth.make_ast_weights(int(cas["s1_seed"]), "wide")
    sd1["classifier.dense.bias"][1] += np.float32(cas["s1_bias_shift"])
    sd2 = synth.make_ast_weights(int(cas["s2_seed"]), "wide")
    sd2["classifier.dense.bias"][1] += np.float32(cas["s2_bias_shift"])
    m1, m2 = str(tmp_path / "s1"), str(tmp_path / "s2")
    _save_model_dir(m1, sd1, float(cas["s1_mean"]), float(cas["s1_std"]), ["Idle", "Swallow"])
    _save_model_dir(m2, sd2, float(cas["s2_mean"]), float(cas["s2_std"]), ["Healthy", "Zenker"])
    wavs = []
    for k in range(2):
        path = str(tmp_path / f"rec{k}.wav")
        pipeline.write_wav_pcm16(path, synth.synth_recording(70 + k, 16000 * 4 + 321), 16000)
        wavs.append(path)
    base = ["--stage1-model-root", m1, "--stage2-model-root", m2, "--file-a", wavs[0], "--file-b", wavs[1],
            "--stage2-threshold", "0.45"]
    plain = pipeline.main(base + ["--output-json", str(tmp_path / "plain.json")])
    cdir = str(tmp_path / "cache")
    first = pipeline.main(base + ["--feature-cache-dir", cdir, "--output-json", str(tmp_path / "c1.json")])
    stores = sorted(glob.glob(os.path.join(cdir, "*.zkc.npz")))
    bundles = sorted(glob.glob(os.path.join(cdir, "*.pt")))
    assert len(stores) == 2 and len(bundles) == 2
    stamp = [os.path.getmtime(p) for p in stores]
    second = pipeline.main(base + ["--feature-cache-dir", cdir, "--output-json", str(tmp_path / "c2.json")])
    assert [os.path.getmtime(p) for p in stores] == stamp                      # served from the cache, nothing rewritten
    for p in stores:
        os.remove(p)                                                            # leave only what the reference writes
    imported = pipeline.main(base + ["--feature-cache-dir", cdir, "--output-json", str(tmp_path / "c3.json")])
    off = pipeline.main(base + ["--feature-cache-dir", cdir, "--disable-cache", "--output-json", str(tmp_path / "c4.json")])
    for run in (first, second, off):
        assert run["per_file"] == plain["per_file"] and run["aggregate"] == plain["aggregate"]
    for k in plain["per_file"]:                # imported bundles: log-mel differs by <= 1 ulp, the counts must still agree
        a, b = plain["per_file"][k], imported["per_file"][k]
        assert {x: a[x] for x in a if "windows" in x} == {x: b[x] for x in b if "windows" in x}
        assert np.abs(np.array(a["stage1_mean_probs"]) - np.array(b["stage1_mean_probs"])).max() <= 1e-5

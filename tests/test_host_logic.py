"""CPU: host-side mirror of the reference script (no GPU calls): window geometry, summaries, aggregation, JSON
schema, WAV reader, config / preprocessor parsing, sharding arithmetic."""
import json
import os

import numpy as np
import pytest
from hypothesis import given, settings
from hypothesis import strategies as st

from oracle import ast_oracle as orc
from zkast import dist as zdist
from zkast import pipeline as pl
from zkast.feature_extraction import ZkASTFeatureExtractor
from zkast.modeling import ZkASTConfig


@pytest.fixture(scope="module")
def cases(golden_dir):
    return json.load(open(os.path.join(golden_dir, "cascade_cases.json")))


def test_window_audio_matches_reference_counts(golden_dir):
    exp = json.load(open(os.path.join(golden_dir, "windows.json")))
    for key, e in exp.items():
        if not key.isdigit():
            continue
        T = int(key)
        n, win, hop = pl.window_geometry(T, 1.0, 0.5)
        assert (n, win, hop) == (e["n"], 16000, 8000)
        if T <= 40000:
            a = np.arange(T, dtype=np.float32)
            wins = pl.window_audio(a, 1.0, 0.5)
            assert len(wins) == e["n"] and all(len(w) == 16000 for w in wins)
    with pytest.raises(ValueError):
        pl.window_geometry(16000, 0.0, 0.5)


@settings(max_examples=60, deadline=None)
@given(st.integers(1, 200000), st.sampled_from([0.25, 0.5, 1.0, 2.0]), st.sampled_from([0.1, 0.25, 0.5, 1.0, 1.5]))
def test_window_geometry_property(T, w, h):
    n, win, hop = pl.window_geometry(T, w, h)
    assert n == len(orc.window_audio(np.zeros(T, np.float32), w, h))
    assert n >= 1 and (T < win or (n - 1) * hop + win <= T)


def test_summaries_match_reference_cases(cases):
    for name, c in cases["cases"].items():
        s1 = np.asarray(c["s1"], np.float32)
        s2 = np.asarray(c["s2_all"], np.float32)
        idx = zdist.gate_indices(s1, c["thr1"], c["min_prob"])
        assert idx.tolist() == c["swallow_idx"], name
        res = [(int(i), s2[i]) for i in idx]
        summ = pl.summarize_stage_outputs(s1, res, ["Idle", "Swallow"], ["Healthy", "Zenker"], c["thr2"],
                                          c["use_argmax"])
        assert json.loads(json.dumps(summ)) == pytest.approx(c["summary"], nan_ok=True, abs=1e-7), name


def test_aggregate_and_schema(cases):
    c = cases["cases"]["thr_0p5"]
    per_file = {"file_0": {"path": "a.wav", **c["summary"]}, "file_1": {"path": "b.wav", **c["summary"]}}
    agg = pl.aggregate_files(per_file, ["a.wav", "b.wav"])
    assert agg["total_windows"] == 2 * c["summary"]["num_windows"]
    assert agg["total_swallow_windows"] == 2 * c["summary"]["stage1_swallow_windows"]
    assert agg["overall_zenker_ratio_over_swallow"] == pytest.approx(
        c["summary"]["stage2_zenker_windows"] / c["summary"]["stage1_swallow_windows"])
    assert set(agg) == {"files_used", "total_windows", "total_idle_windows", "total_swallow_windows",
                        "total_swallow_ratio", "total_swallow_windows_evaluated_stage2", "total_healthy_windows",
                        "total_zenker_windows", "overall_zenker_ratio_over_swallow"}
    none = cases["cases"]["no_swallow"]["summary"]
    agg0 = pl.aggregate_files({"file_0": none, "file_1": none}, ["a", "b"])
    assert agg0["overall_zenker_ratio_over_swallow"] is None and agg0["total_swallow_ratio"] == 0.0


def test_wav_roundtrip_and_formats(tmp_path):
    import struct
    rng = np.random.default_rng(0)
    x = (0.5 * rng.uniform(-1, 1, 4800)).astype(np.float32)
    p = str(tmp_path / "a.wav")
    pl.write_wav_pcm16(p, x, 48000)
    wav, sr = pl.read_wav(p)
    assert sr == 48000 and wav.shape == (1, 4800) and np.abs(wav[0] - x).max() <= 1.0 / 32768
    # stereo float32 with an odd-sized LIST chunk before data
    st2 = np.stack([x, -x], 1).astype("<f4").tobytes()
    hdr = b"RIFF" + struct.pack("<I", 0) + b"WAVE" + b"fmt " + struct.pack("<IHHIIHH", 16, 3, 2, 16000, 128000, 8, 32)
    junk = b"LIST" + struct.pack("<I", 3) + b"abc" + b"\0"
    p2 = str(tmp_path / "b.wav")
    open(p2, "wb").write(hdr + junk + b"data" + struct.pack("<I", len(st2)) + st2)
    wav, sr = pl.read_wav(p2)
    assert sr == 16000 and wav.shape == (2, 4800) and np.array_equal(wav[0], x) and np.array_equal(wav[1], -x)
    # 24-bit PCM
    v = np.round(x * (1 << 23)).astype(np.int32)
    b24 = b"".join(int(i & 0xFFFFFF).to_bytes(3, "little") for i in v)
    hdr = b"RIFF" + struct.pack("<I", 0) + b"WAVE" + b"fmt " + struct.pack("<IHHIIHH", 16, 1, 1, 8000, 24000, 3, 24)
    p3 = str(tmp_path / "c.wav")
    open(p3, "wb").write(hdr + b"data" + struct.pack("<I", len(b24)) + b24)
    wav, sr = pl.read_wav(p3)
    assert sr == 8000 and np.abs(wav[0] - x).max() <= 2.0 ** -23
    with pytest.raises(ValueError):
        open(p3, "wb").write(b"nope")
        pl.read_wav(p3)


def test_discover_two_files(tmp_path):
    d = tmp_path / "Long" / "Zenker" / "006"
    d.mkdir(parents=True)
    for name, n in [("a.wav", 100), ("b.wav", 300), ("c.wav", 200)]:
        pl.write_wav_pcm16(str(d / name), np.zeros(n, np.float32), 16000)
    got = pl.discover_two_files(str(tmp_path), "006", "*.wav")
    assert [os.path.basename(p) for p in got] == ["b.wav", "c.wav"]      # the two longest
    with pytest.raises(ValueError, match="Expected exactly 2 files"):
        pl.discover_two_files(str(tmp_path), "007", "*.wav")


def test_feature_extractor_config_contract(tmp_path):
    fx = ZkASTFeatureExtractor(mean=-1.15, std=3.5)
    d = fx.to_dict()
    json.dumps(d)
    assert d["mean"] == -1.15 and d["std"] == 3.5 and d["num_mel_bins"] == 128 and d["max_length"] == 1024
    fx.save_pretrained(str(tmp_path))
    fx2 = ZkASTFeatureExtractor.from_pretrained(str(tmp_path))
    assert fx2.to_dict() == d and fx.model_input_names[0] == "input_values"
    assert ZkASTFeatureExtractor(mean=0.0, std=1.0).to_dict() != d
    with pytest.raises(ValueError, match="sampling rate"):
        fx._check_rate(8000)
    with pytest.raises(OSError):
        ZkASTFeatureExtractor.from_pretrained(str(tmp_path / "missing"))


def test_config_parsing(tmp_path):
    cfg = {"hidden_size": 768, "num_hidden_layers": 12, "id2label": {"0": "Idle", "1": "Swallow"},
           "label2id": {"Idle": 0, "Swallow": 1}, "architectures": ["ASTForAudioClassification"], "torch_dtype": "float32"}
    json.dump(cfg, open(tmp_path / "config.json", "w"))
    c = ZkASTConfig.from_pretrained(str(tmp_path))
    assert c.num_labels == 2 and c.id2label == {0: "Idle", 1: "Swallow"} and c.layer_norm_eps == 1e-12
    with pytest.raises(ValueError):
        ZkASTConfig(hidden_act="relu")


@settings(max_examples=50, deadline=None)
@given(st.integers(0, 5000), st.integers(1, 8))
def test_shard_ranges_partition(n, world):
    spans = [zdist.shard_range(n, r, world) for r in range(world)]
    assert spans[0][0] == 0 and spans[-1][1] == n
    assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    sizes = [hi - lo for lo, hi in spans]
    assert max(sizes) - min(sizes) <= 1


# ---- batch driver + patient aggregation, against fixtures produced by the reference's own code ----
@pytest.fixture(scope="module")
def BA(golden_dir):
    return json.load(open(os.path.join(golden_dir, "batch_aggregate.json")))


def test_read_ids_and_threshold_resolution(BA, tmp_path):
    from zkast import batch
    p = tmp_path / "test_ids_fold1.txt"
    p.write_text(BA["ids_text"])
    assert batch.read_ids(str(p)) == BA["ids_expected"]
    for fold, exp in BA["thresholds_expected"].items():
        assert batch.resolve_thresholds(BA["threshold_config"], int(fold)) == exp
    assert batch.resolve_thresholds(None, 1) == {}
    assert batch.load_threshold_config(str(tmp_path / "nope.json")) is None


def test_aggregate_matches_reference(BA, tmp_path):
    from zkast import aggregate as agg
    for pid, doc in BA["agg_inputs"].items():
        json.dump(doc, open(tmp_path / f"{pid}_2stage.json", "w"))
    json.dump({"x": 1}, open(tmp_path / "batch_fold1_2stage.json", "w"))
    for thr, exp in BA["agg_expected"].items():
        summ, rows = agg.aggregate(str(tmp_path), float(thr))
        summ.pop("outputs_dir")
        assert summ == exp
        assert len(rows) == exp["num_patient_results"]
    assert agg.infer_ground_truth([]) == "Unknown" and agg.parse_patient_id("/x/006_2stage.json") == "006"


def test_aggregate_cli_writes_csv_and_json(BA, tmp_path, capsys):
    """`python -m zkast.aggregate` flags of utils/aggregate_2stage_results.py:240-266: summary on stdout, --store-output
    default file names inside the outputs directory, explicit --csv / --json paths win, header = the reference's row."""
    import csv
    from zkast import aggregate as agg
    for pid, doc in BA["agg_inputs"].items():
        json.dump(doc, open(tmp_path / f"{pid}_2stage.json", "w"))
    json.dump({"x": 1}, open(tmp_path / "batch_fold1_2stage.json", "w"))      # counted as found, never as a patient
    summ = agg.main(["--outputs-dir", str(tmp_path), "--threshold", "0.3", "--store-output", "--verbose"])
    out = capsys.readouterr().out
    printed = json.loads(out[: out.index("[INFO]")])
    exp = dict(BA["agg_expected"]["0.3"])
    printed.pop("outputs_dir")
    assert printed == exp and summ["threshold"] == 0.3
    rows = list(csv.DictReader(open(tmp_path / "per_patient_results.csv")))
    assert list(rows[0].keys()) == agg.ROW_FIELDS and len(rows) == exp["num_patient_results"]
    doc = json.load(open(tmp_path / "aggregate_summary.json"))
    assert set(doc) == {"summary", "patients"} and len(doc["patients"]) == len(rows)
    assert {k: doc["summary"][k] for k in exp} == exp
    # the written files are not mistaken for patient results on a second run; explicit paths win over the defaults
    c2, j2 = tmp_path / "x.csv", tmp_path / "x.json"
    agg.main(["--outputs-dir", str(tmp_path), "--csv", str(c2), "--json", str(j2)])
    assert c2.exists() and j2.exists()
    assert json.load(open(j2))["summary"]["num_patient_results"] == exp["num_patient_results"]


def test_batch_skip_dry_run_and_error_isolation(tmp_path):
    from zkast import batch
    out = tmp_path / "out"
    out.mkdir()
    (out / "p1_2stage.json").write_text("{}")
    logs = []
    st = batch.run_batch(["p1", "p2"], str(tmp_path), None, None, None, None, str(out), dry_run=True, log=logs.append)
    assert st == {"p1": "skip", "p2": "dry-run"}
    # no files for p2 -> the error is logged and the loop goes on (reference: non-zero exit is logged, loop continues)
    st = batch.run_batch(["p2", "p1"], str(tmp_path), None, None, None, None, str(out), log=logs.append)
    assert st == {"p2": "error", "p1": "skip"} and any("[ERROR] patient p2" in l for l in logs)


def test_feature_cache_keys_match_reference(cases, tmp_path):
    """fingerprint / to_dict equal the real ASTFeatureExtractor's (fixture from transformers 5.15), so `.pt` caches are
    interchangeable with the reference's cached variant."""
    from zkast import cache
    fx = ZkASTFeatureExtractor(mean=-1.1509622, std=3.5340312)
    assert fx.to_dict() == cases["fx_to_dict"]
    assert cache.get_fx_fingerprint(fx) == cases["fx_fingerprint"]
    wav = tmp_path / "rec.wav"
    pl.write_wav_pcm16(str(wav), np.zeros(16000, np.float32), 16000)
    fp = cache.get_fx_fingerprint(fx)
    p1 = cache.build_cache_path(str(tmp_path), str(wav), 1.0, 0.5, 16000, fp)
    assert os.path.basename(p1).startswith("rec_") and p1.endswith(".pt") and len(os.path.basename(p1)) == len("rec_") + 16 + 3
    meta = cache.build_base_metadata(str(wav), 1.0, 0.5, 1, 16000, fp)
    assert meta["audio_size"] == os.path.getsize(wav) and meta["extractor_fingerprint"] == fp
    assert cache.build_cache_path(str(tmp_path), str(wav), 1.0, 0.25, 16000, fp) != p1


def test_prepare_long_audio_layout(tmp_path):
    """utils/PrepareDatasetLongAudio.py: class/specimen/long-subfolder walk, Idle skipped, mono + native rate kept."""
    import struct
    from zkast import dataprep
    raw, out = tmp_path / "raw", tmp_path / "Long"
    x = (0.25 * np.sin(np.arange(4800) / 7.0)).astype(np.float32)
    for cls, spec in [("Healthy", "224_m_60"), ("Zenker", "006_f_71"), ("Idle", "999_x")]:
        d = raw / cls / spec / "Long_recordings"
        d.mkdir(parents=True)
        st2 = np.stack([x, 0.5 * x], 1).astype("<f4").tobytes()          # stereo float32 at 48 kHz
        hdr = b"RIFF" + struct.pack("<I", 0) + b"WAVE" + b"fmt " + struct.pack("<IHHIIHH", 16, 3, 2, 48000, 384000, 8, 32)
        (d / "take1.WAV").write_bytes(hdr + b"data" + struct.pack("<I", len(st2)) + st2)
        (raw / cls / spec / "short").mkdir()
    (raw / "Zenker" / "007_nolong").mkdir()
    logs = []
    assert dataprep.prepare_long_audio(str(raw), str(out), log=logs.append) == 2
    assert sorted(os.listdir(out)) == ["Healthy", "Zenker"] and os.listdir(out / "Healthy") == ["224"]
    wav, sr = pl.read_wav(str(out / "Zenker" / "006" / "take1.wav"))
    assert sr == 48000 and wav.shape == (1, 4800) and np.abs(wav[0] - 0.75 * x).max() <= 1.0 / 32768
    assert any("No long file for specimen: 007_nolong" in str(l) for l in logs)
    assert len(pl.window_audio(wav[0], 0.05, 0.05, 48000)) == 2


def test_evaluate_host_helpers(tmp_path):
    """utils/analyze_ROC_PR_stage1.py:116-160 — split loading with val->test fallback, payload kinds, batching."""
    from zkast import evaluate as ev
    x = np.empty(3, dtype=object)
    x[0] = np.zeros(8000, np.float32); x[1] = {"array": [0.0] * 100, "sampling_rate": 16000}; x[2] = np.ones(5, np.float32)
    np.save(tmp_path / "test_x_fold2.npy", x, allow_pickle=True)
    np.save(tmp_path / "test_y_fold2.npy", np.array([0, 1, 1]))
    X, y, split = ev.load_split(str(tmp_path), 2, "val")
    assert split == "test" and y == [0, 1, 1] and len(X) == 3
    with pytest.raises(FileNotFoundError):
        ev.load_split(str(tmp_path), 3, "val")
    assert [len(b) for b in ev.batched(list(range(19)), 8)] == [8, 8, 3] and list(ev.batched([], 8)) == []
    w = ev.to_waveform(X[1])
    assert w.dtype == np.float32 and w.shape == (100,)
    assert ev.to_waveform(np.arange(4, dtype=np.float64)).dtype == np.float32
    with pytest.raises(ValueError):
        ev.to_waveform({"foo": 1})
    with pytest.raises(TypeError):
        ev.to_waveform(3.5)
    y_pred, cm = ev.evaluate_predictions(np.array([[2.0, 1.0], [0.0, 1.0], [0.5, 0.1]]), [0, 1, 1], 2)
    assert y_pred.tolist() == [0, 1, 0] and cm.tolist() == [[1, 0], [1, 1]]
    p = ev.softmax(np.array([[0.0, 0.0], [1.0, 3.0]], np.float32))
    assert np.allclose(p.sum(1), 1.0) and abs(p[1, 1] - 1 / (1 + np.exp(-2.0))) < 1e-6


def _fake_logmel(n, nf=98, seed=0):
    rng = np.random.default_rng(seed)
    return (rng.normal(-6.0, 3.0, size=(n, nf, 128))).astype(np.float32)


def test_compact_store_roundtrip_and_reference_bundle_interop(tmp_path):
    """compact (N,98,128) store <-> the reference's (N,1024,128) `.pt` bundle.  The bundle written here has
    the reference's keys / metadata (..._cache.py:106-124,181-187) and is accepted back; a store written in this
    build's own format round-trips bit for bit; a foreign or stale entry is ignored, never trusted."""
    import torch
    from zkast import cache
    fx = ZkASTFeatureExtractor(mean=-1.1509622, std=3.5340312)
    wav = tmp_path / "p006_long.wav"
    pl.write_wav_pcm16(str(wav), np.zeros(16000 + 5 * 8000, np.float32), 16000)
    feats = cache.CompactFeatures(_fake_logmel(6), {"note": "unit test"})
    assert cache.n_frames_of(1.0) == 98 and len(feats) == 6 and feats.n_frames == 98
    # expand == HF semantics: zero pad to 1024 rows, then (x - mean) / (2 std) everywhere
    ex = feats.expand(fx)
    assert ex.shape == (6, 1024, 128) and ex.dtype == np.float32
    assert np.array_equal(ex[:, :98], (feats.logmel - np.float32(fx.mean)) / np.float32(2 * fx.std))
    assert np.all(ex[:, 98:] == np.float32(-np.float32(fx.mean)) / np.float32(2 * fx.std))
    raw_fx = ZkASTFeatureExtractor(do_normalize=False)
    assert np.array_equal(feats.expand(raw_fx)[:, :98], feats.logmel) and not feats.expand(raw_fx)[:, 98:].any()
    # inverse: within one ulp of the log-mel value
    back = cache.CompactFeatures.from_expanded(ex, fx, 98)
    assert np.abs(back.logmel - feats.logmel).max() <= 2e-6
    # own format
    logs = []
    store = cache.FeatureCache(str(tmp_path / "c"), log=logs.append)
    key = cache.EntryKey.of(str(wav), 1.0, 0.5, 16000, cache.get_fx_fingerprint(fx))
    assert store.lookup(key, 6, fx) is None
    store.store(key, feats, fx)
    compact_path, bundle_path = store._paths(key)
    assert os.path.exists(compact_path) and os.path.exists(bundle_path)
    assert bundle_path == cache.build_cache_path(str(tmp_path / "c"), str(wav), 1.0, 0.5, 16000, key.fingerprint)
    assert os.path.getsize(compact_path) * 9 < os.path.getsize(bundle_path)
    got = store.lookup(key, 6, fx)
    assert np.array_equal(got.logmel, feats.logmel) and got.provenance == {"note": "unit test"}
    # the `.pt` twin is exactly what the reference writes and reads
    bundle = torch.load(bundle_path)
    assert set(bundle) == {"metadata", "features"} and tuple(bundle["features"].shape) == (6, 1024, 128)
    want = cache.build_base_metadata(str(wav), 1.0, 0.5, 6, 16000, key.fingerprint)
    assert {k: bundle["metadata"][k] for k in want} == want and bundle["metadata"]["feature_shape"] == [6, 1024, 128]
    assert np.array_equal(bundle["features"].numpy(), ex)
    # only the reference bundle present -> imported and compacted
    os.remove(compact_path)
    imp = store.lookup(key, 6, fx)
    assert imp is not None and np.abs(imp.logmel - feats.logmel).max() <= 2e-6 and "imported_from" in imp.provenance
    # another window count / a touched file / a corrupt entry -> not used
    assert store.lookup(key, 7, fx) is None
    open(compact_path, "wb").write(b"not an npz")
    os.remove(bundle_path)
    assert store.lookup(key, 6, fx) is None and any("cannot use" in l for l in logs)
    other = cache.EntryKey.of(str(wav), 1.0, 0.25, 16000, key.fingerprint)
    assert other.digest != key.digest and store.lookup(other, 6, fx) is None
    with pytest.raises(ValueError):
        cache.CompactFeatures(np.zeros((2, 98, 64), np.float32))


def test_entry_key_digest_is_the_reference_key_string(tmp_path):
    """digest = sha256("<abs>|<win>|<hop>|<sr>|<fingerprint>|<size>_<mtime>")[:16] (..._cache.py:97-100)."""
    import hashlib
    from zkast import cache
    wav = tmp_path / "a b.wav"
    pl.write_wav_pcm16(str(wav), np.zeros(100, np.float32), 16000)
    key = cache.EntryKey.of(str(wav), 1.0, 0.5, 16000, "f" * 64)
    st_ = os.stat(wav)
    text = f"{os.path.abspath(wav)}|1.0|0.5|16000|{'f' * 64}|{st_.st_size}_{int(st_.st_mtime)}"
    assert key.digest == hashlib.sha256(text.encode()).hexdigest()[:16] and key.stem == "a b_" + key.digest


def test_window_audio_views_equal_the_oracle_windows():
    rng = np.random.default_rng(3)
    for T, w, h in [(40000, 1.0, 0.5), (16000, 1.0, 0.5), (5000, 1.0, 0.5), (16001, 1.0, 0.5), (80000, 1.0, 1.5),
                    (16000, 0.25, 0.1)]:
        a = rng.standard_normal(T).astype(np.float32)
        got, ref = pl.window_audio(a, w, h), orc.window_audio(a, w, h)
        assert len(got) == len(ref) and all(np.array_equal(g, r) for g, r in zip(got, ref))
    full = pl.window_audio(np.arange(40000, dtype=np.float32), 1.0, 0.5)
    assert full[1].base is not None and full[1][0] == 8000.0          # strided view, not a copy


def test_cache_write_failures_never_stop_the_inference(tmp_path, monkeypatch):
    """torch.save reports a full disk as RuntimeError, the (N,1024,128) expansion of the `.pt` twin can raise
    MemoryError — the reference catches Exception around its cache write (..._cache.py:181-192) and carries on.  store()
    must do the same, write the two files independently and leave no `.part` file behind."""
    from zkast import cache
    fx = ZkASTFeatureExtractor(mean=-1.1509622, std=3.5340312)
    wav = tmp_path / "p007_long.wav"
    pl.write_wav_pcm16(str(wav), np.zeros(16000 + 5 * 8000, np.float32), 16000)
    feats = cache.CompactFeatures(_fake_logmel(6))
    key = cache.EntryKey.of(str(wav), 1.0, 0.5, 16000, cache.get_fx_fingerprint(fx))
    logs = []
    store = cache.FeatureCache(str(tmp_path / "c"), log=logs.append)
    compact_path, bundle_path = store._paths(key)

    def full_disk(*_a, **_k):
        raise RuntimeError("[enforce fail at inline_container.cc] . PytorchStreamWriter failed writing file data/0: file write failed")

    monkeypatch.setattr(cache, "write_reference_bundle", full_disk)
    store.store(key, feats, fx)                                        # must not raise
    assert os.path.exists(compact_path) and not os.path.exists(bundle_path)
    assert any("could not write" in l and "RuntimeError" in l for l in logs)
    os.remove(compact_path)

    monkeypatch.setattr(cache.CompactFeatures, "expand", lambda self, fx: (_ for _ in ()).throw(MemoryError("1.9 GiB")))
    store.store(key, feats, fx)
    assert os.path.exists(compact_path) and any("MemoryError" in l for l in logs)
    os.remove(compact_path)
    monkeypatch.undo()

    real_savez = np.savez

    def failing_savez(f, **kw):
        f.write(b"partial")
        raise OSError(28, "No space left on device")

    monkeypatch.setattr(np, "savez", failing_savez)
    store.store(key, feats, fx)                                        # compact store fails, the twin is still written
    monkeypatch.setattr(np, "savez", real_savez)
    assert not os.path.exists(compact_path) and os.path.exists(bundle_path)
    assert not [f for f in os.listdir(tmp_path / "c") if f.endswith(".part")]
    got = store.lookup(key, 6, fx)                                     # and is usable on the next run
    assert got is not None and np.abs(got.logmel - feats.logmel).max() <= 2e-6


def test_wav_header_reads_headers_only_and_degenerate_formats_count_as_empty(tmp_path):
    """the frame count for discover_two_files comes from the RIFF chunk headers (no sample bytes are read) and
    a header with zero channels / zero bits is an empty recording, not a ZeroDivisionError out of the discovery."""
    import struct
    p = tmp_path / "a.wav"
    pl.write_wav_pcm16(str(p), np.zeros(48000 * 3 + 1, np.float32), 48000)
    tag, ch, sr, bits, nbytes = pl.wav_header(str(p))
    assert (tag, ch, sr, bits, nbytes) == (1, 1, 48000, 16, 2 * (48000 * 3 + 1))
    assert pl._wav_num_frames(str(p)) == 48000 * 3 + 1
    assert pl._length_after_resampling(48000 * 3 + 1, 48000) == 48001 and pl._length_after_resampling(777, 16000) == 777
    assert pl._length_after_resampling(44100, 44100) == 16000
    # same numbers as the full parser
    t2, c2, s2, b2, raw = pl.parse_wav(str(p))
    assert (t2, c2, s2, b2, len(raw)) == (tag, ch, sr, bits, nbytes)
    # a LIST chunk in front of the data chunk, odd-sized (padded) — and a data chunk longer than the file (truncated copy)
    body = struct.pack("<HHIIHH", 1, 2, 16000, 64000, 4, 16)
    blob = (b"RIFF" + struct.pack("<I", 0) + b"WAVE" + b"LIST" + struct.pack("<I", 3) + b"abc\0"
            + b"fmt " + struct.pack("<I", 16) + body + b"data" + struct.pack("<I", 4000) + b"\0" * 400)
    q = tmp_path / "b.wav"
    q.write_bytes(blob)
    assert pl.wav_header(str(q)) == (1, 2, 16000, 16, 400) and pl._wav_num_frames(str(q)) == 100
    # zero channels: counts as length 0
    z = tmp_path / "z.wav"
    z.write_bytes(b"RIFF" + struct.pack("<I", 0) + b"WAVE" + b"fmt " + struct.pack("<I", 16)
                  + struct.pack("<HHIIHH", 1, 0, 16000, 0, 0, 16) + b"data" + struct.pack("<I", 8) + b"\0" * 8)
    assert pl._wav_num_frames(str(z)) == 0
    (tmp_path / "n.wav").write_bytes(b"not a wave file at all")
    assert pl._wav_num_frames(str(tmp_path / "n.wav")) == 0 and pl._wav_num_frames(str(tmp_path / "missing.wav")) == 0


def test_window_audio_returns_plain_slices_like_the_reference():
    """writable views that alias the recording (the reference's `audio[s:s+win]`), not read-only strided views"""
    a = np.arange(40000, dtype=np.float32)
    w = pl.window_audio(a, 1.0, 0.5)
    assert len(w) == 4 and all(x.base is a and x.flags.writeable for x in w)
    w[1][0] = -1.0
    assert a[8000] == -1.0 and w[0][8000] == -1.0


def test_one_feature_store_serves_both_stages_only_when_the_extractors_differ_in_stats_alone():
    a = ZkASTFeatureExtractor(mean=-1.15, std=3.53)
    b = ZkASTFeatureExtractor(mean=-6.5, std=2.75)
    assert pl._extractors_differ_only_in_stats(a, b) and pl._extractors_differ_only_in_stats(a, a)
    c = ZkASTFeatureExtractor(mean=-6.5, std=2.75, do_normalize=False)
    assert not pl._extractors_differ_only_in_stats(a, c)

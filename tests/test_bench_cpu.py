"""bench.py's host-only legs, exercised the way the DRIVER's box runs them (no GPU needed).

Round 3's driver bench died in the CPU-baseline leg: ``oracle/ast_oracle._consts()`` built its two tables lazily, one
after the other, and ``ast_torch_cpu.extract_features_parallel`` was the only caller that reached it COLD from many
threads — every other test had called ``extract_features`` on the main thread first, so no test could see the race.
These tests start a FRESH interpreter for that reason.
"""
import json
import os
import re
import subprocess
import sys
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "zenker-audio-detection_amd"))


def _fresh(code: str, timeout: int = 300) -> str:
    r = subprocess.run([sys.executable, "-c", textwrap.dedent(code)], cwd=ROOT, capture_output=True, text=True,
                       timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    return r.stdout


@pytest.mark.parametrize("attempt", range(4))
def test_parallel_logmel_cold_start_from_many_threads(attempt):
    """64 pool threads hit fbank_frames as the FIRST oracle call of a fresh process, with the interpreter switching
    threads every microsecond (the setting that reproduced the round-3 failure 1 in 8 starts)."""
    out = _fresh("""
        import sys
        sys.setswitchinterval(1e-6)
        sys.path.insert(0, "zenker-audio-detection_amd")
        import numpy as np
        from oracle import ast_torch_cpu as tcpu          # nothing of the oracle has run yet
        rng = np.random.default_rng(3)
        wins = [rng.standard_normal(16000).astype(np.float32) * 0.1 for _ in range(96)]
        f = tcpu.extract_features_parallel(wins, -1.15, 3.53, threads=64)
        from oracle import ast_oracle as orc
        g = orc.extract_features(wins[:4], -1.15, 3.53)
        assert f.shape == (96, 1024, 128) and np.array_equal(f[:4], g)
        print("ok")
    """)
    assert out.strip().endswith("ok")


def test_oracle_has_no_lazily_built_module_state():
    """the constants are import-time, read-only, and published as one object"""
    from oracle import ast_oracle as orc
    mel, hann = orc._consts()
    assert mel.shape == (257, 128) and hann.shape == (400,)
    assert not mel.flags.writeable and not hann.flags.writeable
    assert orc._consts() is orc._CONSTS
    src = open(os.path.join(ROOT, "oracle", "ast_oracle.py")).read()
    assert "global " not in src


def test_time_two_stage_smoke_one_layer():
    """the bench's CPU leg end to end on 2 windows x 1-layer weights: keys, shapes, repeat bookkeeping, the retry form"""
    from oracle import ast_oracle as orc
    from oracle import ast_torch_cpu as tcpu
    from zkast import synth
    sd1 = synth.make_ast_weights(31, "sens", layers=[0])
    sd2 = synth.make_ast_weights(33, "sens", layers=[0])
    rec = synth.synth_recording(100, 16000 + 8000)
    wins = orc.window_audio(rec)
    assert len(wins) == 2
    r = tcpu.time_two_stage(wins, sd1, sd2, (-1.15, 3.53), (-6.5, 2.75), repeats=3, budget_s=120.0)
    assert r["repeats"] == 3 and r["threads"] >= 1
    assert r["logits1"].shape == (2, 2) and r["logits2"].shape == (2, 2)
    assert r["seconds"] > 0 and abs(r["seconds"] - (r["mel_seconds"] + r["forward_seconds"])) < 1e-6
    ref1 = orc.ast_forward(orc.extract_features(wins, -1.15, 3.53), sd1)
    assert np.abs(r["logits1"] - ref1).max() < 1e-4
    one = tcpu.time_two_stage(wins, sd1, sd2, (-1.15, 3.53), (-6.5, 2.75), repeats=3, budget_s=0.0, mel_threads=1)
    assert one["repeats"] == 1                                   # the budget stops the repeats, never the first run
    assert np.array_equal(one["logits1"], r["logits1"]) and np.array_equal(one["logits2"], r["logits2"])


def test_bench_refuses_to_run_without_a_gpu_and_says_so():
    """no CPU fallback: on a box without an MI355X the bench exits non-zero with one clear sentence"""
    r = subprocess.run([sys.executable, "bench.py", "--steps", "1", "--warmup", "0"], cwd=ROOT, capture_output=True,
                       text=True, timeout=300, env={**os.environ, "HIP_VISIBLE_DEVICES": "", "ROCR_VISIBLE_DEVICES": ""})
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    assert r.returncode != 0 and "no CPU fallback" in r.stderr
    assert r.stdout.strip() == ""


def test_bench_leg_guard_records_errors_and_still_prints(tmp_path):
    """the guard bench.py wraps around every auxiliary leg, lifted out of main() by source so that it can run without a
    GPU: a failing leg leaves {"error": ...} under its key, later legs still run, the line is printed exactly once"""
    src = open(os.path.join(ROOT, "bench.py")).read()
    a = src.index('    out["legs"] = {}')
    b = src.index("    single = rank == 0")
    body = textwrap.dedent(src[a:b])
    code = (
        "import json, os, signal, sys, time, traceback\n"
        "class A: time_budget_s = 1000.0\n"
        "args, rank, world, out, t_bench0 = A(), 0, 1, {'value': 1.0}, time.perf_counter()\n"
        "real_stdout = os.dup(1)\n" + body +
        "def boom(): raise TypeError(\"'NoneType' object is not subscriptable\")\n"
        "leg('cpu_baseline', boom)\n"
        "leg('after', lambda: out.__setitem__('after', {'ok': 1}))\n"
        "A.time_budget_s = 0.0\n"
        "leg('late', lambda: out.__setitem__('late', 1), need_s=5)\n"
        "emit(); emit()\n")
    f = tmp_path / "guard.py"
    f.write_text(code)
    r = subprocess.run([sys.executable, str(f)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["value"] == 1.0 and line["after"] == {"ok": 1} and "late" not in line
    assert line["cpu_baseline"]["error"].startswith("TypeError")
    assert line["legs"]["cpu_baseline"].startswith("FAILED") and line["legs"]["after"].startswith("ok")
    assert line["legs"]["late"].startswith("skipped")


def test_bench_constants_are_defined_once_and_match_the_design():
    """round 4: a bad merge duplicated bench.py's header block and the two copies disagreed on the attention pass count"""
    src = open(os.path.join(ROOT, "bench.py")).read()
    for name in ("FLOP_PER_WINDOW_STAGE", "PEAK_F16_DENSE", "PASSES_PER_FLOP", "ATTN_PASSES_PER_FLOP", "TRAFFIC_FILE", "CLOCK_FILE"):
        assert len(re.findall(rf"^{name}\s*=", src, flags=re.M)) == 1, name
    assert src.count("import argparse") == 1
    import importlib
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    assert bench.FLOP_PER_WINDOW_STAGE == 261.03e9 and bench.PEAK_F16_DENSE == 2.5e15
    assert bench.PASSES_PER_FLOP["f16c8"] == 2.0 and bench.PASSES_PER_FLOP["f16x3"] == 3.0 and bench.PASSES_PER_FLOP["f16"] == 1.0
    # attention, per 64-key tile and wave: 512 matrix-pipe cycles for one pass; f16c8 1 024 (QK^T 8 fp16 + 4 fp8 = 512, Vh·P + Vl·P = 512)
    assert bench.ATTN_PASSES_PER_FLOP == {"f16c8": 2.0, "f16x3": 2.5, "f16": 1.0}
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    assert "261.03" in design and "2.5 PFLOP/s" in design

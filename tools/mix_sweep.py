"""Which kernel groups have to run f16x3 for the f16mix mode to keep its margin: stage-1 logit error of a list of
assignments on N windows of the input-sensitive `sens` set, against the fp32 torch-CPU restatement (oracle/ast_torch_cpu.py,
pinned to transformers) or, with --golden, against tests/golden/sens_tail.npz (real transformers, 3 599 windows).
usage: python tools/mix_sweep.py [N=<windows, default 320>] [--golden] 'spec' 'spec' ...   spec = '' (all c8) | '0' | '0:qkv+att,1:mlp' | 'x3'"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "zenker-audio-detection_amd"))
from zkast import ZkASTConfig, ZkASTForAudioClassification, lib, synth  # noqa: E402

S1 = (-1.1509622, 3.5340312)


def parse(spec):
    g = {}
    for item in (v for v in spec.split(",") if v != ""):
        layer, _, groups = item.partition(":")
        g[int(layer)] = tuple(groups.split("+")) if groups else lib.LAYER_GROUPS
    return g


def main():
    args = [a for a in sys.argv[1:] if a != "--golden"]
    golden = "--golden" in sys.argv
    n = 320
    if args and args[0].startswith("N="):
        n = int(args.pop(0)[2:])
    specs = args or ["", "0", "x3"]
    sd = synth.make_ast_weights(31, "sens")
    if golden:
        g = np.load(os.path.join(ROOT, "tests", "golden", "sens_tail.npz"))
        n = int(g["n_windows"])
        rec = synth.synth_recording(int(g["rec_seed"]), 16000 + (n - 1) * 8000)
        ref = g["s1_logits"]
    else:
        from oracle import ast_oracle as orc
        from oracle import ast_torch_cpu as tcpu
        import torch
        rec = synth.synth_recording(11, 16000 + (n - 1) * 8000)
        thr = tcpu.effective_cpus()
        torch.set_num_threads(thr)
        t0 = time.perf_counter()
        ref = tcpu.TorchAST(sd).forward(tcpu.extract_features_parallel(orc.window_audio(rec), *S1, thr))
        print(f"fp32 reference of {n} windows on {thr} threads: {time.perf_counter() - t0:.0f} s", flush=True)
    ctx = lib.get_context(0)
    m = ZkASTForAudioClassification(ZkASTConfig(num_labels=2), sd, stage=0, compute_mode="f16c8", fx_mean=S1[0], fx_std=S1[1])
    ctx.logmel(rec, rec.size, 0, 8000, 16000, n)
    out = {}
    for spec in specs:
        try:
            if spec == "x3":
                m.set_compute_mode("f16x3")
            else:
                m.set_layer_modes(lib.mix_layer_modes(parse(spec)))
        except (lib.ZkError, ValueError) as e:
            print(f"{spec:28s} refused: {e}", flush=True)
            continue
        lg = np.empty((n, 2), np.float32)
        ctx.ast_forward(0, None, None, n, lg)      # (also builds the layer-0 table for this assignment)
        ctx.synchronize()
        t0 = time.perf_counter()
        ctx.ast_forward(0, None, None, n, lg)
        ctx.synchronize()
        dt = time.perf_counter() - t0
        e = np.abs(lg - ref).max(axis=1)
        out[spec or "c8"] = dict(max=float(e.max()), p999=float(np.percentile(e, 99.9)), p99=float(np.percentile(e, 99)),
                                 rms=float(np.sqrt((e ** 2).mean())), median=float(np.median(e)), windows_per_s=n / dt)
        print(f"{spec or 'c8':28s} max {e.max():.2e}  p99.9 {np.percentile(e, 99.9):.2e}  p99 {np.percentile(e, 99):.2e}  rms {np.sqrt((e ** 2).mean()):.2e}  "
              f"median {np.median(e):.2e}   {n / dt:7.1f} windows/s (stage 1 only)", flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump({"windows": n, "reference": "transformers golden" if golden else "torch-CPU fp32 restatement", "results": out},
              open(os.path.join(ROOT, "gpurun_out", "mix_sweep.json"), "w"), indent=1)


if __name__ == "__main__":
    main()

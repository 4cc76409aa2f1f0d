#!/usr/bin/env python
"""bench.py — BASELINE.json metric on MI355X: 1-s audio windows/s through the full two-stage cascade
(log-mel + AST stage 1 + gate + AST stage 2), one process per GPU.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (config.workload): BASELINE.json configs[2] — batch of 1024 synthetic 1 s / 0.5 s-hop windows per GPU, cut on
the device from one resident 16 kHz recording; gate rate g = 1.0 (every window goes through BOTH stages: "AST x2",
the worst case of SURVEY.md §8d).  A step = log-mel of the 1024 windows + stage-1 forward + gate selection +
stage-2 forward on the selected windows (+ the RCCL all-gather of both logit tables when N > 1).
Weights are the synthetic "wide" set (no checkpoint exists offline); data and weights are stated in the JSON line.
The headline `value` is measured in a compute mode that meets the 1e-3 logit tolerance: f16c8 (fp16 MFMA pass + one fp8
pass carrying the split corrections; ~1.5e-4 measured).  The 3-pass f16x3 mode (same tolerance) and the single-pass
fp16 rate (fails the tolerance) are reported beside it with their measured logit differences, never as `value`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "zenker-audio-detection_amd"))
sys.path.insert(0, ROOT)

FLOP_PER_WINDOW_STAGE = 261.03e9          # dense AST forward at S=1214
PEAK_F16_DENSE = 2.5e15                   # MI355X_MICROARCH.md: BF16/FP16 MFMA dense peak
S, D, F = 1214, 768, 3072
GEMM_SHAPES = {"gemm_qkv": (3 * D, D), "gemm_o": (D, D), "gemm_fc1": (F, D), "gemm_fc2": (D, F)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=1024, help="windows per GPU per step")
    ap.add_argument("--gate-rate", type=float, default=1.0, help="fraction of windows forced through stage 2")
    ap.add_argument("--micro-batch", type=int, default=0, help="0 = library default (auto)")
    ap.add_argument("--mode", default="f16c8", choices=["f16c8", "f16x3", "f16"])
    ap.add_argument("--no-fast", action="store_true", help="skip the secondary f16x3 / single-pass fp16 measurements")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--cpu-windows", type=int, default=6)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from zkast import ZkASTConfig, ZkASTFeatureExtractor, ZkASTForAudioClassification, lib, synth

    ctx = lib.get_context(local_rank)
    ctx.set_micro_batch(args.micro_batch)
    S1 = (-1.1509622, 3.5340312)
    S2 = (-6.5, 2.75)
    sd1 = synth.make_ast_weights(21, "wide")
    sd2 = synth.make_ast_weights(22, "wide")
    m1 = ZkASTForAudioClassification(ZkASTConfig(num_labels=2), sd1, stage=0, compute_mode=args.mode,
                                     device=local_rank, fx_mean=S1[0], fx_std=S1[1])
    m2 = ZkASTForAudioClassification(ZkASTConfig(num_labels=2), sd2, stage=1, compute_mode=args.mode,
                                     device=local_rank, fx_mean=S2[0], fx_std=S2[1])
    B = args.batch
    hop, win = 8000, 16000
    n_samples = win + (B - 1) * hop
    rec = synth.synth_recording(100 + rank, n_samples)
    audio = torch.from_numpy(rec).to(dev)                       # inputs resident in HBM before the timed region
    s1_logits = torch.empty((B, 2), dtype=torch.float32, device=dev)
    s2_logits = torch.empty((B, 2), dtype=torch.float32, device=dev)
    K = max(1, int(round(args.gate_rate * B)))
    if world > 1:
        g1 = torch.empty((world * B, 2), dtype=torch.float32, device=dev)
        g2 = torch.empty((world * B, 2), dtype=torch.float32, device=dev)

    def step():
        ctx.logmel(audio, n_samples, 0, hop, win, B)
        ctx.ast_forward(0, None, None, B, s1_logits)
        # gate selection: the top-g fraction of stage-1 p_swallow, ascending window order (SURVEY.md §8d knob)
        p_sw = torch.softmax(s1_logits, dim=1)[:, 1]
        idx = torch.sort(torch.topk(p_sw, K).indices).values.to(torch.int32).contiguous()
        torch.cuda.current_stream().synchronize()
        ctx.ast_forward(1, None, idx, K, s2_logits)
        if world > 1:
            dist.all_gather_into_tensor(g1, s1_logits)
            dist.all_gather_into_tensor(g2, s2_logits)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.synchronize()

    def timed(steps, warmup, profile=False):
        for _ in range(warmup):
            step()
        barrier()
        if profile:
            ctx.prof_begin()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        barrier()
        dt = time.perf_counter() - t0
        prof = ctx.prof_end() if profile else None
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, prof

    dt, prof = timed(args.steps, args.warmup, profile=True)
    windows = world * B * args.steps
    value = windows / dt

    # ---- per-kernel roofline from the HIP-event timings of the timed region (rank 0) ----
    roof_all = {}
    for name in ("gemm_qkv", "gemm_o", "gemm_fc1", "gemm_fc2", "gemm_patch", "attention"):
        ms, cnt, fl = prof[name]                               # fl = algorithmic FLOPs EXECUTED (exact pruning applied)
        if cnt:
            roof_all[name] = dict(ms_per_launch=ms / cnt, launches=cnt, tflops=fl / (ms * 1e-3) / 1e12,
                                  gflop_per_launch=fl / cnt / 1e9, share=ms / (dt * 1e3))
    for name in ("layernorm", "logmel", "embed", "head"):
        ms, cnt, _ = prof[name]
        if cnt:
            roof_all[name] = dict(ms_per_launch=ms / cnt, launches=cnt, share=ms / (dt * 1e3))
    executed = sum(prof[n][2] for n in ("gemm_qkv", "gemm_o", "gemm_fc1", "gemm_fc2", "gemm_patch", "attention"))
    dom = max((k for k in roof_all if "tflops" in roof_all[k]), key=lambda k: roof_all[k]["share"])
    roofline = {"kernel": dom, "bound": "mfma", "achieved": roof_all[dom]["tflops"], "peak": PEAK_F16_DENSE / 1e12,
                "unit": "TFLOP/s", "frac": roof_all[dom]["tflops"] * 1e12 / PEAK_F16_DENSE, "traffic": None,
                "ms_per_launch": roof_all[dom]["ms_per_launch"], "launches": roof_all[dom]["launches"]}
    # HBM traffic of the dominant kernel: PMC counters cannot be read inside this process; the value is the one
    # measured with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, gfx950 FETCH correction applied) and
    # committed under profiles/ (same kernel, micro-batch 107)
    try:
        tr_file = "r01_f_pmc_traffic.json" if args.mode == "f16c8" else "r01_b_pmc_traffic.json"
        tr = json.load(open(os.path.join(ROOT, "profiles", tr_file)))["kernels"]
        key = {"gemm_fc1": "gemm_fc1(gelu)", "gemm_qkv": "gemm_qkv(store)", "gemm_fc2": "gemm_resid(o,fc2)",
               "gemm_o": "gemm_resid(o,fc2)", "attention": "attention"}[dom]
        roofline["traffic"] = tr[key]["hbm_bytes_per_launch_corrected"]
        roofline["traffic_source"] = f"profiles/{tr_file} (rocprofv3 --pmc, separate passes)"
    except Exception:
        pass
    e2e_flops = value / world * (1.0 + K / B) * FLOP_PER_WINDOW_STAGE
    out = {
        "metric": "1-s audio windows/sec two-stage (mel+ASTx2)", "value": value, "unit": "windows/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.mode, "data": "synthetic",
        "config": {"workload": "configs[2]: full two-stage cascade, batch=1024 1-s windows per GPU, hop 0.5 s, "
                               f"gate rate g={K / B:.2f}", "windows_per_gpu": B, "stage2_windows_per_gpu": K,
                   "micro_batch": args.micro_batch, "weights": "synthetic splitmix64 'wide' set (seeds 21/22)",
                   "parallelism": f"window-sharded x{world}, RCCL all-gather of logits" if world > 1 else "single GPU"},
        "roofline": roofline,
        "roofline_end_to_end": {"algorithmic_gflop_per_window_stage": FLOP_PER_WINDOW_STAGE / 1e9,
                                "achieved_tflops_per_gpu": e2e_flops / 1e12, "frac_of_f16_dense_peak": e2e_flops / PEAK_F16_DENSE,
                                "executed_gflop_per_window_stage": executed / (args.steps * (B + K)) / 1e9,
                                "executed_tflops_per_gpu": executed / dt / 1e12,
                                "note": "executed = exact last-layer pruning applied (tokens 0/1 only feed the head); "
                                        "f16c8 issues 2 (f16x3: 3) matrix-pipe passes per GEMM FLOP counted here once"},
        "kernels": roof_all,
    }

    # ---- secondary: the 3-pass mode (same tolerance) and single-pass fp16 (fails the 1e-3 tolerance), each with its
    #      measured stage-1 logit difference to the headline mode ----
    if not args.no_fast and args.mode == "f16c8":
        ref1 = s1_logits.cpu().numpy().copy()
        nsec = max(1, args.steps // 2)
        for key, mode, note in (("x3_mode", "f16x3", "(hi,lo) fp16 pairs, 3 MFMA passes; also meets the 1e-3 logit tolerance"),
                                ("fast_mode", "f16", "single fp16 MFMA pass; exceeds the 1e-3 logit tolerance, not the headline")):
            m1.set_compute_mode(mode)
            m2.set_compute_mode(mode)
            dtf, _ = timed(nsec, 1)
            err = float(np.abs(s1_logits.cpu().numpy() - ref1).max())
            out[key] = {"dtype": mode, "value": world * B * nsec / dtf, "unit": "windows/s",
                        "stage1_logit_max_abs_diff_vs_f16c8": err, "note": note}
        m1.set_compute_mode(args.mode)
        m2.set_compute_mode(args.mode)

    # ---- BASELINE.json configs[1] as an extra line: batch 256, stage-1 only (log-mel + forward), both modes ----
    if rank == 0 and world == 1:
        def stage1_b256(reps=3):
            n256 = min(256, B)
            ctx.logmel(audio, n_samples, 0, hop, win, n256); ctx.ast_forward(0, None, None, n256, s1_logits)
            barrier(); t0 = time.perf_counter()
            for _ in range(reps):
                ctx.logmel(audio, n_samples, 0, hop, win, n256)
                ctx.ast_forward(0, None, None, n256, s1_logits)
            barrier()
            return n256 * reps / (time.perf_counter() - t0)
        cfg1 = {"workload": "configs[1]: batch=256 windows, stage-1 only (log-mel + AST forward)", "unit": "windows/s"}
        for md in ("f16c8", "f16x3", "f16"):
            m1.set_compute_mode(md); cfg1[md] = stage1_b256()
        m1.set_compute_mode(args.mode)
        out["config1_stage1_b256"] = cfg1

    # ---- CPU baseline: the oracle (numpy restatement) on this box's host cores, rank 0, N=1 only ----
    if rank == 0 and world == 1 and not args.no_cpu:
        from oracle import ast_oracle as orc
        n_cpu = args.cpu_windows
        wins = orc.window_audio(rec[: win + (n_cpu - 1) * hop])
        W1, W2 = orc.ASTWeights(sd1), orc.ASTWeights(sd2)
        try:                                    # report the BLAS threads actually used, not the box's core count
            from threadpoolctl import threadpool_info
            blas_threads = max([int(i.get("num_threads", 1)) for i in threadpool_info()] or [1])
        except Exception:
            blas_threads = None
        allowed = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()
        t0 = time.perf_counter()
        f1 = orc.extract_features(wins, *S1)
        l1 = orc.ast_forward(f1, W1)
        f2 = orc.extract_features(wins, *S2)
        l2 = orc.ast_forward(f2, W2)
        tc = time.perf_counter() - t0
        del f2
        out["cpu_baseline"] = {"value": n_cpu / tc, "unit": "windows/s",
                               "cores": blas_threads if blas_threads else allowed, "kind": "port",
                               "sample": f"{n_cpu} windows x (log-mel + AST) x 2 stages, numpy/BLAS fp32 oracle, "
                                         f"{tc:.1f} s wall; BLAS threads {blas_threads}, cpus allowed {allowed}, "
                                         f"os.cpu_count {os.cpu_count()} (log-mel part is single-threaded numpy)"}
        out["parity_in_bench"] = {"stage1_logit_max_abs_err_vs_oracle": None}
        # re-run the headline mode once so the comparison is against its logits
        ctx.logmel(audio, n_samples, 0, hop, win, B)
        ctx.ast_forward(0, None, None, B, s1_logits)
        out["parity_in_bench"]["stage1_logit_max_abs_err_vs_oracle"] = float(
            np.abs(s1_logits.cpu().numpy()[:n_cpu] - l1).max())
        del l2

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""BASELINE.json configs[3] and configs[4] as runnable workloads (synthetic audio and weights), one process per GPU:

    python tools/run_configs.py --config 3                          # 30-min 48 kHz recording, window-sharded cascade
    python tools/run_configs.py --config 4 [--patients 64]          # patients x 2 files x 5 min, patient-sharded batch
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29501 \
        tools/run_configs.py --config 4

config 3 is timed end to end from FILE BYTES: each rank uploads, decodes and resamples (on the device) only the slice of
the file its windows need, runs stage 1, the logits are all-gathered through the C ABI (RCCL), every rank derives the
gate, the gated windows are re-partitioned, stage 2, second all-gather.  config 4 writes <pid>_2stage.json per patient
exactly as the single-process driver does and ends with the patient-level confusion matrix of
utils/aggregate_2stage_results.py.  Rank 0 prints one JSON line per config.
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "zenker-audio-detection_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, required=True, choices=[3, 4])
    ap.add_argument("--patients", type=int, default=64)
    ap.add_argument("--minutes", type=float, default=None, help="recording length (default 30 for config 3, 5 per file for config 4)")
    ap.add_argument("--mode", default=None, help="compute mode (default: zkast.lib.DEFAULT_COMPUTE_MODE)")
    args = ap.parse_args()
    if args.mode is None:
        from zkast import lib as _zl
        args.mode = _zl.DEFAULT_COMPUTE_MODE
    # one JSON line on stdout: gloo / RCCL print banners there, so fd 1 goes to stderr and the line to the saved descriptor
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        os.write(real_stdout, (json.dumps(obj) + "\n").encode())

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    device = 0 if os.environ.get("ZK_BENCH_ONE_GPU") else int(os.environ.get("LOCAL_RANK", "0"))
    from zkast import ZkASTConfig, ZkASTFeatureExtractor, ZkASTForAudioClassification, aggregate, batch, lib, pipeline, synth
    from zkast import dist as zdist
    ctx = lib.get_context(device)
    gather = None
    rccl_error = None
    if world > 1:
        import torch
        import torch.distributed as tdist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        tdist.init_process_group("gloo", rank=rank, world_size=world)
        try:
            zdist.init_comm(ctx, rank, world)
        except lib.ZkError as e:      # e.g. a rehearsal with several ranks on one GPU, which RCCL refuses
            rccl_error = str(e)
            print(f"[run_configs] rank {rank}: RCCL communicator failed ({e})", file=sys.stderr)
        ok = torch.tensor([1 if ctx.comm_info()[1] == world else 0])      # every rank takes the same path (MIN over ranks)
        tdist.all_reduce(ok, op=tdist.ReduceOp.MIN)
        if not bool(ok.item()):
            if ctx.comm_info()[1] == world:
                ctx.comm_destroy()
            if os.environ.get("ZK_BENCH_REQUIRE_RCCL") == "1":
                raise SystemExit(f"[run_configs] rank {rank}: ZK_BENCH_REQUIRE_RCCL=1 and no RCCL communicator over "
                                 f"{world} ranks ({rccl_error or 'a peer failed'})")
            print(f"[run_configs] rank {rank}: host gathers over gloo instead", file=sys.stderr)

            def gather(b):
                box = [None] * world
                tdist.all_gather_object(box, b)
                return box
        else:
            gather = ctx.allgather_bytes
    rccl_world = ctx.comm_info()[1] if (world > 1 and gather == ctx.allgather_bytes) else (1 if world == 1 else 0)
    self_desc = {"rccl_world": rccl_world, "collective_fallback": bool(world > 1 and rccl_world != world)}
    S1, S2 = (-1.1509622, 3.5340312), (-6.5, 2.75)
    cas = np.load(os.path.join(ROOT, "tests", "golden", "cascade.npz"))
    sd1 = synth.make_ast_weights(21, "wide")
    sd1["classifier.dense.bias"][1] += np.float32(cas["s1_bias_shift"])      # centres the random gate (≈ half the windows)
    sd2 = synth.make_ast_weights(22, "wide")
    sd2["classifier.dense.bias"][1] += np.float32(cas["s2_bias_shift"])
    m1 = ZkASTForAudioClassification(ZkASTConfig(num_labels=2), sd1, stage=0, compute_mode=args.mode, device=device,
                                     fx_mean=S1[0], fx_std=S1[1])
    m2 = ZkASTForAudioClassification(ZkASTConfig(num_labels=2), sd2, stage=1, compute_mode=args.mode, device=device,
                                     fx_mean=S2[0], fx_std=S2[1])
    fx1, fx2 = ZkASTFeatureExtractor(mean=S1[0], std=S1[1]), ZkASTFeatureExtractor(mean=S2[0], std=S2[1])

    def barrier():
        ctx.synchronize()
        if gather:
            gather(b"\0")

    if args.config == 3:
        minutes = args.minutes or 30.0
        x48 = synth.synth_recording(7, int(minutes * 60 * 48000))
        pcm = np.round(np.clip(x48, -1, 1 - 1 / 32768) * 32768).astype("<i2").tobytes()
        src = zdist.WavSource(pcm, 1, 16, 1, 48000)
        casc = zdist.ZkShardedCascade(m1, fx1, m2, fx2, rank, world)
        casc(zdist.WavSource(pcm[: 48000 * 2 * 20], 1, 16, 1, 48000))      # warm-up on 20 s
        barrier()
        t0 = time.perf_counter()
        s1, idx, s2 = casc(src)
        barrier()
        dt = time.perf_counter() - t0
        mine = json.dumps({"rank": rank, "bytes_uploaded": int(casc.h2d_samples), "slices_uploaded": int(casc.uploads),
                           "gather_s": round(casc.stats.get("gather_s", 0.0), 6)}).encode()
        per_rank = [json.loads(b.decode()) for b in gather(mine)] if gather else [json.loads(mine.decode())]
        if rank == 0:
            emit(({**self_desc, "per_rank": per_rank, "config": "configs[3]: %.0f-min 48 kHz PCM16 recording, 1 s / 0.5 s-hop windows sharded over %d GPU(s)"
                              % (minutes, world), "n_gpus": world, "windows": int(casc.n_windows), "gated_windows": int(len(idx)),
                              "seconds_end_to_end": dt, "windows_per_s": casc.n_windows / dt,
                              "file_bytes": len(pcm), "bytes_uploaded_by_rank0": int(casc.h2d_samples),
                              "collective": "zk_allgather_logits (RCCL)" if casc.comm_ctx is not None else
                              ("none (one rank)" if world == 1 else "host gather over gloo (RCCL unavailable)"),
                              "dtype": args.mode, "data": "synthetic"}))
    else:
        minutes = args.minutes or 5.0
        tmp = tempfile.mkdtemp(prefix="zk_cfg4_") if rank == 0 else None
        if gather:
            tmp = [b for b in gather((tmp or "").encode()) if b][0].decode()
        ids = []
        if rank == 0:
            base = [synth.synth_recording(50 + k, int(minutes * 60 * 16000)) for k in range(4)]
            for p in range(args.patients):
                cls = "Zenker" if p % 2 else "Healthy"
                d = os.path.join(tmp, "Long", cls, f"{p:03d}")
                os.makedirs(d)
                for k in range(2):
                    pipeline.write_wav_pcm16(os.path.join(d, f"rec{k}.wav"), base[(p + k) % 4] * (0.5 + 0.1 * (p % 5)), 16000)
        ids = [f"{p:03d}" for p in range(args.patients)]
        barrier()
        t0 = time.perf_counter()
        summ = {}
        st = batch.run_batch(ids, os.path.join(tmp, "Long"), m1, fx1, m2, fx2, os.path.join(tmp, "out"),
                             log=lambda *_: None, rank=rank, world=world, gather_bytes=gather, summaries=summ)
        barrier()
        dt = time.perf_counter() - t0
        if rank == 0:
            summary, rows = aggregate.aggregate(os.path.join(tmp, "out"), 0.5)
            nwin = sum(v["total_windows"] for v in summ.values())
            emit(({**self_desc, "config": "configs[4]: %d synthetic patients x 2 files x %.0f min @16 kHz, patient-sharded over %d GPU(s), "
                              "in-process batch driver + patient-level aggregation" % (args.patients, minutes, world),
                              "n_gpus": world, "patients_ok": sum(v == "ok" for v in st.values()), "windows": int(nwin),
                              "seconds_end_to_end": dt, "windows_per_s": nwin / dt,
                              "confusion_matrix": summary["confusion_matrix"], "dtype": args.mode, "data": "synthetic"}))
        barrier()
        if rank == 0:
            shutil.rmtree(tmp, ignore_errors=True)
    if world > 1:
        if ctx.comm_info()[1] == world:
            ctx.comm_destroy()
        tdist.destroy_process_group()


if __name__ == "__main__":
    main()

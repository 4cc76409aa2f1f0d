"""In-process batch driver: the reference launches one Python subprocess per patient and re-reads both 86 M-parameter
models each time (src/run_batch_simple_2stage.py:273-291).  Here both stages are loaded once, stay resident on the GPU
and every patient is a call of `run_patient`.  Same inputs (test_ids_fold<k>.txt, threshold-config JSON, output
directory, skip-if-exists / --force) and the same per-patient `<pid>_2stage.json`.
"""
from __future__ import annotations

import argparse
import json
import os
from typing import Any, Dict, List, Optional

from . import pipeline as pl
from .lib import DEFAULT_COMPUTE_MODE


def read_ids(ids_path: str) -> List[str]:
    """Patient ids of a `test_ids_fold<k>.txt` file: one entry per non-blank line, the id is the last `/`-separated
    component of the entry (src/run_batch_simple_2stage.py:48-57)."""
    with open(ids_path) as f:
        entries = (ln.strip() for ln in f)
        return [e.rsplit("/", 1)[-1] for e in entries if e]


def load_threshold_config(config_path: Optional[str]) -> Optional[dict]:
    """The threshold-config JSON, or None when no path was given or the file does not exist (:60-65)."""
    if config_path and os.path.isfile(config_path):
        with open(config_path) as f:
            return json.load(f)
    return None


def resolve_thresholds(threshold_config: Optional[dict], fold: int) -> Dict[str, float]:
    """The --stage1-threshold / --stage2-threshold flags build_cmd would append (:96-118): per-fold block
    {"folds": {"<k>": {"stage1": {"threshold": t}, "stage2": {...}}}} (utils/extract_thresholds_per_fold.py:93-122)
    first, else the single {"thresholds": {...}} format; absent entries keep the script defaults (0.5)."""
    out: Dict[str, float] = {}
    if not threshold_config:
        return out
    folds = threshold_config.get("folds", {})
    key = str(fold)
    src = folds[key] if folds and key in folds else threshold_config.get("thresholds", {})
    if "stage1" in src:
        out["stage1_threshold"] = src["stage1"]["threshold"]
    if "stage2" in src:
        out["stage2_threshold"] = src["stage2"]["threshold"]
    return out


def shard_patients(patients: List[str], rank: int, world: int) -> List[str]:
    """Patients are the independent unit of the batch ("shard by patient/file first"): rank r takes every
    world-th patient of the list, which spreads long and short recordings of a sorted id list evenly."""
    return list(patients[rank::world])


def run_batch(patients: List[str], long_audio_root: str, model_s1, fx_s1, model_s2, fx_s2, output_dir: str = "outputs",
              pattern: str = "*.wav", force: bool = False, dry_run: bool = False, log=print, rank: int = 0,
              world: int = 1, gather_bytes=None, summaries: Optional[Dict[str, Any]] = None,
              **opts: Any) -> Dict[str, str]:
    """The patient loop of main() (:258-292).  opts: window_sec, hop_sec, stage1_threshold, stage2_threshold,
    stage1_forward_min_prob, stage2_argmax, stage1_model_root, stage2_model_root.  Returns {pid: status}.

    world > 1: this rank runs `shard_patients(patients, rank, world)` on its own GPU (the models stay resident, the
    per-patient JSON files go to the shared output directory exactly as in the single-process run) and the statuses —
    and, when `summaries` is given, each patient's "aggregate" block, which is all utils/aggregate_2stage_results.py
    reads (:119-129) — are gathered at the end, so every rank returns the full table."""
    os.makedirs(output_dir, exist_ok=True)
    status = {}
    local_summ: Dict[str, Any] = {}
    for pid in shard_patients(patients, rank, world):
        expected_json = os.path.join(output_dir, f"{pid}_2stage.json")
        if os.path.exists(expected_json) and not force:
            log(f"[SKIP] {pid} (exists: {expected_json})")
            status[pid] = "skip"
            continue
        log(f"[RUN] {pid}")
        if dry_run:
            status[pid] = "dry-run"
            continue
        try:
            files = pl.discover_two_files(long_audio_root, pid, pattern)
            output = pl.run_patient(files, model_s1, fx_s1, model_s2, fx_s2, dict(opts))
            with open(expected_json, "w") as f:
                json.dump(output, f, indent=2)
            local_summ[pid] = output.get("aggregate")
            log(f"[DONE] {pid} OK")
            status[pid] = "ok"
        except Exception as e:  # one bad patient must not stop the batch (:286-289)
            log(f"[ERROR] patient {pid}: {type(e).__name__}: {e}")
            status[pid] = "error"
    if world > 1 and gather_bytes is not None:
        status_all, summ_all = {}, {}
        for blob in gather_bytes(json.dumps({"status": status, "summ": local_summ}).encode()):
            d = json.loads(blob.decode())
            status_all.update(d["status"])
            summ_all.update(d["summ"])
        status = {pid: status_all[pid] for pid in patients if pid in status_all}      # list order, as one process reports it
        local_summ = summ_all
    if summaries is not None:
        summaries.update(local_summ)
    log("Batch complete.")
    return status


def build_arg_parser():
    ap = argparse.ArgumentParser(description="In-process batch launcher for two-stage window inference (MI355X).")
    ap.add_argument("--fold", type=int, required=True)
    ap.add_argument("--ids-root", default=None)
    ap.add_argument("--long-audio-root", required=True)
    ap.add_argument("--pattern", default="*.wav")
    ap.add_argument("--window-sec", type=float, default=1.0)
    ap.add_argument("--hop-sec", type=float, default=0.5)
    ap.add_argument("--output-dir")
    ap.add_argument("--threshold-config")
    ap.add_argument("--stage1-model-root")
    ap.add_argument("--stage2-model-root")
    ap.add_argument("--stage1-forward-min-prob", type=float)
    ap.add_argument("--stage2-argmax", action="store_true")
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--dry-run", action="store_true")
    ap.add_argument("--compute-mode", default=DEFAULT_COMPUTE_MODE, choices=["f16", "f16c8", "f16x3", "f16mix"])
    return ap


def main(argv=None):
    args = build_arg_parser().parse_args(argv)
    ids_root = args.ids_root or os.path.join(os.getcwd(), "data_ast_stage2")
    ids_path = os.path.join(ids_root, f"test_ids_fold{args.fold}.txt")
    if not os.path.exists(ids_path):
        raise FileNotFoundError(f"IDs file not found: {ids_path}")
    patients = read_ids(ids_path)
    if not patients:
        print("No patient IDs found; exiting.")
        return {}
    print(f"Read {len(patients)} patient IDs from {ids_path}")
    s1 = args.stage1_model_root or os.path.join(os.getcwd(), "runs", "ast_classifier_stage1", f"fold{args.fold}", "best")
    s2 = args.stage2_model_root or os.path.join(os.getcwd(), "runs", "ast_classifier_stage2", f"fold{args.fold}", "best")
    opts = dict(window_sec=args.window_sec, hop_sec=args.hop_sec, stage1_model_root=s1, stage2_model_root=s2,
                stage1_forward_min_prob=args.stage1_forward_min_prob, stage2_argmax=args.stage2_argmax)
    opts.update(resolve_thresholds(load_threshold_config(args.threshold_config), args.fold))
    if args.dry_run:
        return run_batch(patients, args.long_audio_root, None, None, None, None, args.output_dir or "outputs",
                         args.pattern, args.force, True, **opts)
    # one process per GPU (python -m torch.distributed.run --nproc-per-node N -m zkast.batch ...): the patient list is
    # sharded across the ranks, each rank keeps both models resident on its own GPU; torch.distributed (gloo) only
    # ships the RCCL unique id, the final gather of the per-patient records runs through the C ABI
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    device = int(os.environ.get("LOCAL_RANK", "0"))
    fx_s1, model_s1 = pl.load_stage_model(s1, ["Idle", "Swallow"], 0, args.compute_mode, device)
    fx_s2, model_s2 = pl.load_stage_model(s2, ["Healthy", "Zenker"], 1, args.compute_mode, device)
    gather = None
    if world > 1:
        import torch.distributed as tdist
        from . import dist as zdist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        tdist.init_process_group("gloo", rank=rank, world_size=world)
        zdist.init_comm(model_s1._ctx, rank, world)
        gather = model_s1._ctx.allgather_bytes
    try:
        return run_batch(patients, args.long_audio_root, model_s1, fx_s1, model_s2, fx_s2,
                         args.output_dir or "outputs", args.pattern, args.force, False,
                         log=print if rank == 0 else (lambda *_: None), rank=rank, world=world, gather_bytes=gather,
                         **opts)
    finally:
        if world > 1:
            model_s1._ctx.comm_destroy()
            tdist.destroy_process_group()


if __name__ == "__main__":
    main()

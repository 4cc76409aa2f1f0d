"""Patient-level aggregation of the per-patient `<pid>_2stage.json` files — utils/aggregate_2stage_results.py:63-191,
same summary keys.  Host-side; consumes exactly what pipeline.run_patient / batch.run_batch write."""
from __future__ import annotations

import glob
import json
import os
from typing import Dict, List, Optional, Tuple


_CLASS_DIRS = ("Healthy", "Zenker")            # ground truth is the class folder in the recording's path
_SUFFIX = "_2stage.json"
# (ground truth, prediction) -> confusion-matrix cell; "Zenker" is the positive class
_CELL = {("Zenker", "Zenker"): "tp", ("Healthy", "Healthy"): "tn", ("Healthy", "Zenker"): "fp", ("Zenker", "Healthy"): "fn"}


def infer_ground_truth(files_used: List[str]) -> str:
    """Class of a patient = the `/Healthy/` or `/Zenker/` folder (any case) in the path of the FIRST recording used;
    "Unknown" without recordings or without such a folder (utils/aggregate_2stage_results.py:63-72)."""
    first = files_used[0].lower() if files_used else ""
    return next((c for c in _CLASS_DIRS if f"/{c.lower()}/" in first), "Unknown")


def classify_result(gt: str, ratio: Optional[float], threshold: float) -> Tuple[Optional[str], Dict[str, int]]:
    """(predicted label, one-hot confusion cell) of one patient: Zenker when the zenker-over-swallow ratio reaches the
    threshold; no prediction (all cells 0) without a ratio or without ground truth (:75-89)."""
    cells = dict.fromkeys(("tp", "tn", "fp", "fn"), 0)
    if ratio is None or gt not in _CLASS_DIRS:
        return None, cells
    pred = _CLASS_DIRS[int(ratio >= threshold)]
    cells[_CELL[(gt, pred)]] = 1
    return pred, cells


def parse_patient_id(filename: str) -> str:
    """`<pid>_2stage.json` -> `<pid>`; any other file name -> its stem."""
    name = os.path.basename(filename)
    return name[: -len(_SUFFIX)] if name.endswith(_SUFFIX) else os.path.splitext(name)[0]


def aggregate(outputs_dir: str, threshold: float = 0.5):
    """:100-191 -> (summary dict, per-patient rows)."""
    files = sorted(glob.glob(os.path.join(outputs_dir, "*_2stage.json")))
    rows = []
    skipped_no_ratio = skipped_unknown_gt = 0
    for path in files:
        if os.path.basename(path).startswith("batch_fold"):
            continue
        try:
            with open(path, "r") as f:
                data = json.load(f)
        except Exception:
            continue
        agg = data.get("aggregate", {})
        ratio = agg.get("overall_zenker_ratio_over_swallow")
        gt = infer_ground_truth(agg.get("files_used") or [])
        pred, cm = classify_result(gt, ratio, threshold)
        skipped_no_ratio += ratio is None
        skipped_unknown_gt += gt == "Unknown"
        rows.append({"patient_id": parse_patient_id(path), "gt": gt, "ratio": ratio, "predicted_label": pred, **cm,
                     "swallow_windows": agg.get("total_swallow_windows"),
                     "zenker_windows": agg.get("total_zenker_windows"),
                     "healthy_windows": agg.get("total_healthy_windows"),
                     "total_windows": agg.get("total_windows"), "json_path": path})
    tp, tn, fp, fn = (sum(r[k] for r in rows) for k in ("tp", "tn", "fp", "fn"))
    evaluated = tp + tn + fp + fn
    precision = tp / (tp + fp) if (tp + fp) else None
    recall = tp / (tp + fn) if (tp + fn) else None
    specificity = tn / (tn + fp) if (tn + fp) else None
    f1 = (2 * precision * recall / (precision + recall)
          if (precision is not None and recall is not None and (precision + recall) > 0) else None)
    balanced = (((recall or 0.0) + (specificity or 0.0)) / 2
                if (recall is not None and specificity is not None) else None)
    summary = {
        "outputs_dir": outputs_dir,
        "threshold": threshold,
        "num_files_found": len(files),
        "num_patient_results": len(rows),
        "skipped_no_ratio": int(skipped_no_ratio),
        "skipped_unknown_gt": int(skipped_unknown_gt),
        "confusion_matrix": {"TP": tp, "TN": tn, "FP": fp, "FN": fn},
        "metrics": {"accuracy": (tp + tn) / evaluated if evaluated else 0.0, "precision": precision,
                    "recall_sensitivity": recall, "specificity": specificity, "f1": f1,
                    "balanced_accuracy": balanced},
    }
    return summary, rows


ROW_FIELDS = ["patient_id", "gt", "ratio", "predicted_label", "tp", "tn", "fp", "fn", "swallow_windows",
              "zenker_windows", "healthy_windows", "total_windows", "json_path"]


def write_outputs(summary: dict, rows: List[dict], csv_path: Optional[str] = None, json_path: Optional[str] = None) -> None:
    """The two optional artefacts of utils/aggregate_2stage_results.py:196-237: a per-patient CSV (header =
    ROW_FIELDS, one row per patient JSON) and a {"summary", "patients"} JSON."""
    import csv
    if csv_path:
        with open(csv_path, "w", newline="") as cf:
            w = csv.DictWriter(cf, fieldnames=ROW_FIELDS)
            w.writeheader()
            for r in rows:
                w.writerow({k: r.get(k) for k in ROW_FIELDS})
    if json_path:
        with open(json_path, "w") as jf:
            json.dump({"summary": summary, "patients": [{k: r.get(k) for k in ROW_FIELDS} for r in rows]}, jf, indent=2)


def build_arg_parser():
    """Same flags as utils/aggregate_2stage_results.py:240-266."""
    import argparse
    ap = argparse.ArgumentParser(description="Aggregate two-stage per-patient inference JSON outputs.")
    ap.add_argument("--outputs-dir", default="outputs", help="Directory containing *_2stage.json files.")
    ap.add_argument("--threshold", type=float, default=0.5, help="Zenker ratio threshold for positive prediction.")
    ap.add_argument("--csv", help="Optional CSV path for per-patient rows.")
    ap.add_argument("--json", help="Optional JSON path for full summary + per-patient data.")
    ap.add_argument("--verbose", action="store_true", help="Verbose logging.")
    ap.add_argument("--store-output", action="store_true",
                    help="Store json and csv with default name in output folder.")
    return ap


def main(argv=None):
    """`python -m zkast.aggregate --outputs-dir <dir> [--threshold t] [--csv p] [--json p] [--store-output]`:
    prints the summary as JSON (:194); --store-output writes per_patient_results.csv and aggregate_summary.json
    into the outputs directory (:197-199), --csv / --json name the files explicitly and win over the defaults."""
    args = build_arg_parser().parse_args(argv)
    summary, rows = aggregate(args.outputs_dir, args.threshold)
    print(json.dumps(summary, indent=2))
    csv_path = args.csv or (os.path.join(args.outputs_dir, "per_patient_results.csv") if args.store_output else None)
    json_path = args.json or (os.path.join(args.outputs_dir, "aggregate_summary.json") if args.store_output else None)
    write_outputs(summary, rows, csv_path, json_path)
    if args.verbose:
        if csv_path:
            print(f"[INFO] Wrote per-patient CSV: {csv_path}")
        if json_path:
            print(f"[INFO] Wrote aggregate JSON: {json_path}")
    return summary


if __name__ == "__main__":  # pragma: no cover
    main()

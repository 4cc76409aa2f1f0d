"""Patient-level aggregation of the per-patient `<pid>_2stage.json` files — utils/aggregate_2stage_results.py:63-191,
same summary keys.  Host-side; consumes exactly what pipeline.run_patient / batch.run_batch write."""
from __future__ import annotations

import glob
import json
import os
from typing import Dict, List, Optional, Tuple


def infer_ground_truth(files_used: List[str]) -> str:
    """:63-72: ground truth from the path of the first file."""
    if not files_used:
        return "Unknown"
    lower = files_used[0].lower()
    if "/healthy/" in lower:
        return "Healthy"
    if "/zenker/" in lower:
        return "Zenker"
    return "Unknown"


def classify_result(gt: str, ratio: Optional[float], threshold: float) -> Tuple[Optional[str], Dict[str, int]]:
    """:75-89."""
    cm = {"tp": 0, "tn": 0, "fp": 0, "fn": 0}
    if ratio is None or gt == "Unknown":
        return None, cm
    pred = "Zenker" if ratio >= threshold else "Healthy"
    key = {("Healthy", "Healthy"): "tn", ("Healthy", "Zenker"): "fp", ("Zenker", "Zenker"): "tp",
           ("Zenker", "Healthy"): "fn"}.get((gt, pred))
    if key:
        cm[key] = 1
    return pred, cm


def parse_patient_id(filename: str) -> str:
    base = os.path.basename(filename)
    if base.endswith("_2stage.json"):
        return base[: -len("_2stage.json")]
    return os.path.splitext(base)[0]


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

"""Front / profile / rear frame counts from temporal position (reference `head_recon/eval_head_coverage.py:12-28`)."""
from __future__ import annotations

import argparse
import json
from pathlib import Path


def coverage_bucket(progress: float) -> str:
    """Same thresholds as `validation_reporting._bucket` (`validation_reporting.py:40-45`)."""
    if progress < 0.20 or progress > 0.80:
        return "front"
    return "profile" if 0.35 <= progress <= 0.65 else "rear"


def evaluate_head_coverage(n_frames: int) -> dict:
    report = {"front": 0, "profile": 0, "rear": 0, "n_frames": max(n_frames, 0)}
    for i in range(max(n_frames, 0)):
        report[coverage_bucket(i / max(1, n_frames - 1))] += 1
    return report


def main():
    ap = argparse.ArgumentParser(description="Compute coarse head coverage report.")
    ap.add_argument("--transforms", required=True, type=Path)
    ap.add_argument("--output", type=Path, default=Path("02_Visual_Engine/output/head_recon/head_coverage.json"))
    a = ap.parse_args()
    with open(a.transforms, "r", encoding="utf-8") as f:
        n = len(json.load(f).get("frames", []))
    a.output.parent.mkdir(parents=True, exist_ok=True)
    with open(a.output, "w", encoding="utf-8") as f:
        json.dump(evaluate_head_coverage(n), f, indent=2)
    print(f"[head_recon] Wrote coverage report: {a.output}")


if __name__ == "__main__":
    main()

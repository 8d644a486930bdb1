"""validation_reporting -- PSNR / global SSIM of `renders/` vs `gt/` on the deterministic frame
subset, bucketed front / profile / rear (drop-in for `02_Visual_Engine/validation_reporting.py`;
SURVEY.md §8f-2).  CPU reporting; reads PNGs with the engine's own codec (no PIL needed)."""
from __future__ import annotations

import argparse
import json
import math
from pathlib import Path

import numpy as np

from .engine.io_formats import load_image_rgb
from .head_recon.eval_head_coverage import coverage_bucket as _bucket

_CHECKLIST = """# Human Review Checklist

- [ ] Jawline continuity in profile views.
- [ ] Ear geometry plausibility in left/right profile.
- [ ] Neck-head transition remains stable across motion.
- [ ] No visible shimmer/flicker in slow turns.
- [ ] Maxilla/mandible changes remain anatomically plausible.
"""


def psnr(a: np.ndarray, b: np.ndarray) -> float:
    """8-bit PSNR, 99.0 for identical images (reference :16-20)."""
    mse = float(np.mean((a - b) ** 2))
    return 99.0 if mse == 0.0 else 20.0 * math.log10(255.0 / math.sqrt(mse))


def _luma(x: np.ndarray) -> np.ndarray:
    if x.ndim == 3:
        x = 0.299 * x[:, :, 0] + 0.587 * x[:, :, 1] + 0.114 * x[:, :, 2]
    return x.astype(np.float64)


def ssim_global(a: np.ndarray, b: np.ndarray) -> float:
    """Single-window SSIM over the whole luma image, C1=(0.01*255)^2, C2=(0.03*255)^2 (reference :23-37)."""
    x, y = _luma(a), _luma(b)
    mx, my = x.mean(), y.mean()
    vx, vy, cov = ((x - mx) ** 2).mean(), ((y - my) ** 2).mean(), ((x - mx) * (y - my)).mean()
    c1, c2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    return float(((2 * mx * my + c1) * (2 * cov + c2)) / ((mx * mx + my * my + c1) * (vx + vy + c2)))


def _find_latest_train_dir(model_path: Path) -> Path:
    train_dir = model_path / "train"
    if not train_dir.exists():
        raise FileNotFoundError(f"Missing train directory: {train_dir}")
    runs = [p for p in train_dir.iterdir() if p.is_dir() and p.name.startswith("ours_")]
    if not runs:
        raise FileNotFoundError(f"No ours_* directories in {train_dir}")
    return max(runs, key=lambda p: int(p.name.split("_")[-1]))


def generate_report(model_path: Path, deterministic_frames_dir: Path, output_dir: Path):
    latest = _find_latest_train_dir(Path(model_path))
    renders_dir, gt_dir = latest / "renders", latest / "gt"
    if not renders_dir.exists() or not gt_dir.exists():
        raise FileNotFoundError(f"Missing renders/gt directories in {latest}")
    manifest = Path(deterministic_frames_dir) / "deterministic_indices_manifest.json"
    if not manifest.exists():
        raise FileNotFoundError(f"Missing deterministic manifest: {manifest}")
    with open(manifest, "r", encoding="utf-8") as f:
        rows = json.load(f).get("exports", [])
    max_index = max((int(r.get("index", 0)) for r in rows), default=1)
    metrics = []
    for row in rows:
        idx, name = int(row["index"]), row["source"]
        if not (renders_dir / name).exists() or not (gt_dir / name).exists():
            continue
        a = load_image_rgb(renders_dir / name).astype(np.float32)
        b = load_image_rgb(gt_dir / name).astype(np.float32)
        progress = idx / max(1, max_index)
        metrics.append({"index": idx, "frame": name, "progress": progress, "bucket": _bucket(progress),
                        "psnr": psnr(a, b), "ssim": ssim_global(a, b)})
    summary = {"count": len(metrics), "by_bucket": {}}
    for bucket in ("front", "profile", "rear"):
        sel = [m for m in metrics if m["bucket"] == bucket]
        summary["by_bucket"][bucket] = ({"count": len(sel), "psnr": float(np.mean([m["psnr"] for m in sel])),
                                         "ssim": float(np.mean([m["ssim"] for m in sel]))} if sel
                                        else {"count": 0, "psnr": None, "ssim": None})
    output_dir = Path(output_dir)
    output_dir.mkdir(parents=True, exist_ok=True)
    scores_path = output_dir / "strict_scores.json"
    with open(scores_path, "w", encoding="utf-8") as f:
        json.dump({"summary": summary, "rows": metrics}, f, indent=2)
    checklist_path = output_dir / "human_review_checklist.md"
    checklist_path.write_text(_CHECKLIST, encoding="utf-8")
    print(f"[validation_reporting] Wrote strict report: {scores_path}")
    print(f"[validation_reporting] Wrote checklist: {checklist_path}")


def main():
    ap = argparse.ArgumentParser(description="Generate deterministic validation report.")
    ap.add_argument("--model_path", required=True, type=Path)
    ap.add_argument("--deterministic_frames_dir", required=True, type=Path)
    ap.add_argument("--output_dir", type=Path, default=Path("02_Visual_Engine/output/model/eval_strict/reports"))
    a = ap.parse_args()
    generate_report(a.model_path, a.deterministic_frames_dir, a.output_dir)


if __name__ == "__main__":
    main()

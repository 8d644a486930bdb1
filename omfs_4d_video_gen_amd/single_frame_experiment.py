"""single_frame_experiment -- one-command regression in the spirit of
`02_Visual_Engine/single_frame_experiment.py`: take frame 0 of a dataset, train on it
through `train_ghost.train`, render it back through `render_surgery` with zero surgical offsets and save
ground truth and render side by side for a look / a PSNR.

Differences from the reference script, on purpose: paths are arguments instead of hard-coded Windows
paths (`single_frame_experiment.py:22-29`), and `run` repeats the single frame `--copies` times (default
50) because `train_ghost.run_quality_gates` refuses datasets with fewer than 50 training frames
(`train_ghost.py:109-112`) -- the reference script trips over that gate.  `build_single_frame_dataset(src, dst)`
itself (one copy) writes exactly what the reference's function writes (golden: `tests/golden/scenarios.py`).
"""
from __future__ import annotations

import argparse
import json
import shutil
from pathlib import Path

import numpy as np

from . import render_surgery, train_ghost
from .engine.io_formats import load_image_rgb, write_png
from .validation_reporting import psnr


def build_single_frame_dataset(data_dir: Path, out_dir: Path, copies: int = 1) -> Path:
    """Dataset holding frame 0 of `data_dir` only.  With `copies == 1` this is the reference's function
    (`single_frame_experiment.py:32-81`, golden-pinned: same files, same three transforms files byte for byte, same batched
    npz, same message) with its two hard-coded directories as arguments; `copies > 1` appends repeats of the frame (same
    image, FLAME parameters and camera, timestep_index i) so that the set passes `train_ghost.run_quality_gates`."""
    data_dir, out_dir = Path(data_dir), Path(out_dir)
    if out_dir.exists():
        shutil.rmtree(out_dir)
    for sub in ("images", "flame_param", "fg_masks"):
        (out_dir / sub).mkdir(parents=True, exist_ok=True)
    has_mask = (data_dir / "fg_masks" / "00000_00.png").exists()      # the reference requires it; datasets without mattes train too
    shutil.copy2(data_dir / "images" / "00000_00.png", out_dir / "images" / "00000_00.png")
    shutil.copy2(data_dir / "flame_param" / "00000.npz", out_dir / "flame_param" / "00000.npz")
    if has_mask:
        shutil.copy2(data_dir / "fg_masks" / "00000_00.png", out_dir / "fg_masks" / "00000_00.png")
    else:
        shutil.rmtree(out_dir / "fg_masks")
    with open(data_dir / "transforms_train.json") as f:
        full = json.load(f)
    frame0 = full["frames"][0]
    frames = [frame0]
    for i in range(1, copies):
        name = f"{i:05d}_00.png"
        shutil.copy2(out_dir / "images" / "00000_00.png", out_dir / "images" / name)
        shutil.copy2(out_dir / "flame_param" / "00000.npz", out_dir / "flame_param" / f"{i:05d}.npz")
        fr = {**frame0, "file_path": f"images/{name}", "flame_param_path": f"flame_param/{i:05d}.npz", "timestep_index": i}
        if has_mask:
            shutil.copy2(out_dir / "fg_masks" / "00000_00.png", out_dir / "fg_masks" / name)
            if "fg_mask_path" in frame0:
                fr["fg_mask_path"] = f"fg_masks/{name}"
        frames.append(fr)
    top = {k: full[k] for k in ("camera_angle_x", "camera_angle_y", "fl_x", "fl_y", "cx", "cy", "w", "h")}
    for name in ("transforms_train.json", "transforms_test.json", "transforms_val.json"):
        with open(out_dir / name, "w") as f:
            json.dump({**top, "frames": frames if name == "transforms_train.json" else frames[:1]}, f, indent=2)
    p0 = dict(np.load(data_dir / "flame_param" / "00000.npz", allow_pickle=True))
    batched = {}
    for k, v in p0.items():
        if v.ndim == 1:
            batched[k] = v
        else:
            row = v[None, ...] if v.shape[0] != 1 else v
            batched[k] = row if copies == 1 or k == "static_offset" else np.repeat(row, copies, 0)
    np.savez(out_dir / "flame_param.npz", **batched)
    shutil.copy2(data_dir / "canonical_flame_param.npz", out_dir / "canonical_flame_param.npz")
    print(f"[single_frame] Built {out_dir} (1 frame)" if copies == 1 else f"[single_frame] Built {out_dir} ({copies} copies of frame 0)")
    return out_dir


def run(data_dir: str, work_dir: str, iterations: int = 3000, copies: int = 50) -> dict:
    work = Path(work_dir)
    ds = build_single_frame_dataset(Path(data_dir), work / "data_single_frame", copies)
    model = work / "model_single_frame"
    train_ghost.train(str(ds), str(model), iterations=iterations, resolution=-1)
    mod = render_surgery.create_modified_dataset(str(ds), 0.0, 0.0)
    try:
        renders = Path(render_surgery.render_with_gaussians(str(model), mod))
    finally:
        shutil.rmtree(mod, ignore_errors=True)
    gt = load_image_rgb(ds / "images" / "00000_00.png")
    out = load_image_rgb(renders / "00000.png")
    write_png(work / "single_frame_gt.png", gt)
    write_png(work / "single_frame_render.png", out)
    score = psnr(out.astype(np.float32), gt.astype(np.float32))
    print(f"[single_frame] GT: {work / 'single_frame_gt.png'}  render: {work / 'single_frame_render.png'}  PSNR {score:.2f} dB")
    return {"psnr": score, "gt": str(work / "single_frame_gt.png"), "render": str(work / "single_frame_render.png")}


def main():
    ap = argparse.ArgumentParser(description="Single-frame train / render / compare experiment.")
    ap.add_argument("--data_dir", required=True)
    ap.add_argument("--work_dir", required=True)
    ap.add_argument("--iterations", type=int, default=3000)
    ap.add_argument("--copies", type=int, default=50)
    a = ap.parse_args()
    run(a.data_dir, a.work_dir, a.iterations, a.copies)


if __name__ == "__main__":
    main()

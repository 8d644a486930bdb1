"""preprocess_video -- the dataset-format half of `02_Visual_Engine/preprocess_video.py`.

`convert_to_gaussianavatars_format(vhap_export_dir, output_dir)` (reference `:200-426`) turns a VHAP
export (transforms.json + images/ + fg_masks/ + flame_param/*.npz) into the directory layout the
engine trains on: batched `flame_param.npz`, all-zero `canonical_flame_param.npz` (T = 1), the 90/10
train/test split with val = test, per-frame intrinsics, and the combined `transforms.json`.
The VHAP tracker itself (`run_vhap_*`, reference `:110-197`) is a third-party tool that is not part of
this repository: `preprocess_with_vhap` only runs when a `vhap_repo` checkout is supplied, exactly as in
the reference, and otherwise says so.  No GPU work happens here.
"""
from __future__ import annotations

import argparse
import json
import math
import shutil
from pathlib import Path

import numpy as np

N_FLAME_VERTS = 5143
_PER_FRAME_KEYS = ("expr", "rotation", "neck_pose", "jaw_pose", "eyes_pose", "translation")
_DEFAULT_TRANSFORM = [[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 1], [0, 0, 0, 1]]


def _replace_tree(src: Path, dst: Path, label: str, pattern: str) -> None:
    if dst.exists():
        shutil.rmtree(dst)
    if src.exists():
        shutil.copytree(src, dst)
        print(f"[convert] Copied {label}: {len(list(dst.glob(pattern)))} files")


def _first_row(a: np.ndarray, batched_ndim: int) -> np.ndarray:
    """Per-frame VHAP arrays come as (1, ...) or (...): drop the leading frame axis if present."""
    return a[0] if a.ndim > batched_ndim - 1 else a


def convert_to_gaussianavatars_format(vhap_export_dir: Path, output_dir: Path) -> dict:
    vhap_export_dir, output_dir = Path(vhap_export_dir), Path(output_dir)
    output_dir.mkdir(parents=True, exist_ok=True)
    transforms_path = vhap_export_dir / "transforms.json"
    if not transforms_path.exists():
        raise FileNotFoundError(f"VHAP export not found: {transforms_path}")
    with open(transforms_path, "r") as f:
        frames = json.load(f).get("frames", [])
    if not frames:
        raise ValueError("No frames in VHAP export")
    print(f"[convert] Converting {len(frames)} frames to GaussianAvatars format")

    # intrinsics come from the FIRST FRAME (the export's top-level values are normalised differently)
    f0 = frames[0]
    fl_x = f0.get("fl_x", 1000.0)
    fl_y = f0.get("fl_y", fl_x)
    cx, cy = f0.get("cx", 480.0), f0.get("cy", 270.0)
    w, h = f0.get("w", 960), f0.get("h", 540)
    fov_x = f0.get("camera_angle_x", 2 * math.atan(w / (2 * fl_x)))
    fov_y = f0.get("camera_angle_y", 2 * math.atan(h / (2 * fl_y)))
    print("[convert] Camera intrinsics from first frame:")
    print(f"  fl_x={fl_x:.1f}, fl_y={fl_y:.1f}")
    print(f"  cx={cx:.1f}, cy={cy:.1f}")
    print(f"  w={w}, h={h}")
    print(f"  fov_x={math.degrees(fov_x):.1f} deg")

    _replace_tree(vhap_export_dir / "images", output_dir / "images", "images", "*.png")
    _replace_tree(vhap_export_dir / "fg_masks", output_dir / "fg_masks", "fg_masks", "*.png")
    _replace_tree(vhap_export_dir / "flame_param", output_dir / "flame_param", "flame_param", "*.npz")

    # batched FLAME parameters over the frames whose per-frame file exists
    rows = {k: [] for k in _PER_FRAME_KEYS}
    shape = static_offset = None
    dynamic = []
    for i, frame in enumerate(frames):
        t = frame.get("timestep_index", i)
        src = vhap_export_dir / frame.get("flame_param_path", f"flame_param/{t:05d}.npz")
        if not src.exists():
            continue
        p = dict(np.load(src))
        if shape is None:
            shape = p["shape"]
        for k in _PER_FRAME_KEYS:
            rows[k].append(_first_row(p[k], 2))
        if "static_offset" in p and static_offset is None:
            static_offset = p["static_offset"]
        if "dynamic_offset" in p:
            dynamic.append(_first_row(p["dynamic_offset"], 3))
    T = len(rows["expr"])
    width = {"expr": 100, "rotation": 3, "neck_pose": 3, "jaw_pose": 3, "eyes_pose": 6, "translation": 3}
    batched = {"shape": shape if shape is not None else np.zeros(300, dtype=np.float32)}
    for k in _PER_FRAME_KEYS:
        batched[k] = np.stack(rows[k]) if rows[k] else np.zeros((T, width[k]), dtype=np.float32)
    batched["static_offset"] = static_offset if static_offset is not None else np.zeros((1, N_FLAME_VERTS, 3), dtype=np.float32)
    batched["dynamic_offset"] = np.stack(dynamic) if dynamic else np.zeros((T, N_FLAME_VERTS, 3), dtype=np.float32)
    np.savez(output_dir / "flame_param.npz", **batched)
    print(f"[convert] Saved batched FLAME params: {output_dir / 'flame_param.npz'}")
    print(f"  Shape: {batched['shape'].shape}")
    print(f"  Expr:  {batched['expr'].shape}")
    print(f"  Translation range: [{batched['translation'].min():.3f}, {batched['translation'].max():.3f}]")

    # neutral pose: its presence selects the FLAME-rigged (dynamic) loader downstream
    canonical = {"shape": batched["shape"], "expr": np.zeros((1, batched["expr"].shape[1]), dtype=np.float32)}
    for k, n in (("rotation", 3), ("neck_pose", 3), ("jaw_pose", 3), ("eyes_pose", 6), ("translation", 3)):
        canonical[k] = np.zeros((1, n), dtype=np.float32)
    canonical["static_offset"] = batched["static_offset"]
    canonical["dynamic_offset"] = np.zeros((1, N_FLAME_VERTS, 3), dtype=np.float32)
    np.savez(output_dir / "canonical_flame_param.npz", **canonical)
    print(f"[convert] Saved canonical FLAME params: {output_dir / 'canonical_flame_param.npz'}")

    entries = []
    for i, frame in enumerate(frames):
        e = {"file_path": frame.get("file_path", f"images/{i:05d}_00.png"),
             "flame_param_path": frame.get("flame_param_path", f"flame_param/{i:05d}.npz"),
             "transform_matrix": frame.get("transform_matrix", _DEFAULT_TRANSFORM),
             "timestep_index": frame.get("timestep_index", i),
             "camera_index": frame.get("camera_index", 0),
             "camera_angle_x": frame.get("camera_angle_x", fov_x),
             "w": frame.get("w", w), "h": frame.get("h", h)}
        if frame.get("fg_mask_path"):
            e["fg_mask_path"] = frame["fg_mask_path"]
        entries.append(e)
    top = {"camera_angle_x": fov_x, "camera_angle_y": fov_y, "fl_x": fl_x, "fl_y": fl_y, "cx": cx, "cy": cy, "w": w, "h": h,
           "frames": entries, "timestep_indices": list(range(T)), "camera_indices": [0]}
    split = max(1, T - T // 10)
    for name, part in (("transforms_train.json", entries[:split]), ("transforms_test.json", entries[split:]),
                       ("transforms_val.json", entries[split:]), ("transforms.json", entries)):
        with open(output_dir / name, "w") as f:
            json.dump({**top, "frames": part}, f, indent=2)
    print("[convert] Created transforms JSON files")
    print(f"  Train frames: {len(entries[:split])}")
    print(f"  Test frames:  {len(entries[split:])}")
    return {"num_frames": T, "image_size": (w, h), "output_dir": str(output_dir)}


def preprocess_with_vhap(video_path: str, output_dir: str, **kwargs) -> dict:
    """The reference drives the third-party VHAP tracker here (`:429-513`); this repository ships no tracker."""
    raise NotImplementedError("VHAP tracking is a third-party tool (vhap_repo) and is not part of this engine; "
                              "run it as the reference does, then call convert_to_gaussianavatars_format()")


def main():
    ap = argparse.ArgumentParser(description="Preprocess video with VHAP for GaussianAvatars.")
    ap.add_argument("--video", type=str, default=None, help="Path to input video (required unless --convert-only).")
    ap.add_argument("--output_dir", type=str, default="02_Visual_Engine/data", help="Output directory.")
    ap.add_argument("--convert-only", action="store_true", help="Only convert existing VHAP export to GaussianAvatars format.")
    ap.add_argument("--vhap_export_dir", type=str, default=None, help="VHAP export folder (required with --convert-only).")
    a, _ = ap.parse_known_args()
    if a.convert_only:
        if not a.vhap_export_dir:
            ap.error("--vhap_export_dir is required with --convert-only")
        convert_to_gaussianavatars_format(Path(a.vhap_export_dir), Path(a.output_dir))
    else:
        preprocess_with_vhap(a.video, a.output_dir)


if __name__ == "__main__":
    main()

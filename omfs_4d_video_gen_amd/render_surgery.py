"""render_surgery -- drop-in for `02_Visual_Engine/render_surgery.py` of the reference.

Same call surface (reference `render_surgery.py:35-541`; the names its tests import,
`test/test_render_surgery.py:11-17`): mm -> FLAME offsets, a temporary copy of the dataset with
edited FLAME parameters, a child process running `<ENGINE_DIR>/render.py` with the reference's
argv (`:289-301`), discovery of `<model>/train/ours_<iter>/renders`, deterministic frame export
and the ffmpeg stitch.  ENGINE_DIR is this package's `engine/` (MI355X HIP kernels) instead of the
un-vendored CUDA checkout; override with OMFS_ENGINE_DIR.

    python -m omfs_4d_video_gen_amd.render_surgery --lefort_mm 3 --bsso_mm 5 --model_path M --data_dir D
"""
from __future__ import annotations

import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
from pathlib import Path
from typing import Any

import numpy as np

SCALE_FACTOR = 0.001  # mm -> FLAME internal units (reference :35)
REPO_DIR = Path(os.environ.get("OMFS_ENGINE_DIR", Path(__file__).resolve().parent / "engine"))
RENDER_SCRIPT = REPO_DIR / "render.py"
_SPLIT_FILES = ("transforms_train.json", "transforms_test.json", "transforms_val.json")


def compute_offset(input_mm: float, sensitivity: float) -> float:
    """Clinical millimetres -> FLAME-space offset (reference :40-42)."""
    return input_mm * sensitivity * SCALE_FACTOR


def _get_ffmpeg_path() -> str:
    try:
        import imageio_ffmpeg
        return imageio_ffmpeg.get_ffmpeg_exe()
    except ImportError:
        pass
    found = shutil.which("ffmpeg")
    if found:
        return found
    raise FileNotFoundError("ffmpeg not found. Install via: pip install imageio-ffmpeg")


def load_deformation_map(path: str | None) -> dict[str, Any]:
    """Optional JSON with translation_axis / jaw_axis / lefort_scale / bsso_scale (reference :60-71)."""
    if not path:
        return {}
    where = Path(path)
    if not where.exists():
        raise FileNotFoundError(f"Deformation map not found: {where}")
    with open(where, "r", encoding="utf-8") as fh:
        content = json.load(fh)
    if not isinstance(content, dict):
        raise ValueError("Deformation map JSON must contain an object at the top level.")
    return content


def choose_rig_mode(requested_mode: str, canonical_head_asset: str | None) -> tuple[str, str]:
    """(effective mode, reason); hybrid falls back to flame_only without an asset (reference :74-85)."""
    if requested_mode == "flame_only":
        return "flame_only", "explicitly requested"
    if canonical_head_asset and Path(canonical_head_asset).exists():
        return "hybrid_full_head", "canonical head asset found"
    return "flame_only", "hybrid requested but canonical head asset missing"


def _shift_axis(arr: np.ndarray, axis: int, delta: float) -> np.ndarray:
    """Add delta to column `axis` of a (3,) vector or a (T,3) table; returns a copy."""
    out = arr.copy()
    if out.ndim == 1:
        out[axis] += delta
    else:
        out[:, axis] += delta
    return out


def modify_flame_params(source_npz: str, output_npz: str, lefort_offset: float, bsso_offset: float,
                        deformation_map: dict[str, Any] | None = None) -> None:
    """Le Fort I -> translation[..., axis 1]; BSSO -> jaw_pose[..., axis 0]; every other key is
    carried over unchanged (reference :88-141)."""
    data = dict(np.load(source_npz, allow_pickle=True))
    dm = deformation_map or {}
    if "translation" in data:
        data["translation"] = _shift_axis(data["translation"], int(dm.get("translation_axis", 1)),
                                          lefort_offset * float(dm.get("lefort_scale", 1.0)))
    if "jaw_pose" in data:
        data["jaw_pose"] = _shift_axis(data["jaw_pose"], int(dm.get("jaw_axis", 0)),
                                       bsso_offset * float(dm.get("bsso_scale", 1.0)))
    np.savez(output_npz, **data)


def create_modified_dataset(data_dir: str, lefort_offset: float, bsso_offset: float,
                            deformation_map: dict[str, Any] | None = None) -> str:
    """Temporary dataset: images linked, every FLAME npz edited, transforms re-pointed to the
    per-frame files (reference :144-242).  The caller owns and removes the directory."""
    temp_dir = tempfile.mkdtemp(prefix="surgical_render_")
    src, dst = Path(data_dir), Path(temp_dir)

    try:
        os.symlink(os.path.abspath(src / "images"), dst / "images", target_is_directory=True)
    except (OSError, NotImplementedError):
        shutil.copytree(src / "images", dst / "images")

    per_frame = src / "flame_param"          # VHAP's name is singular
    if per_frame.is_dir():
        (dst / "flame_param").mkdir(exist_ok=True)
        for name in os.listdir(per_frame):
            if name.endswith(".npz"):
                modify_flame_params(str(per_frame / name), str(dst / "flame_param" / name), lefort_offset, bsso_offset,
                                    deformation_map=deformation_map)
    if (src / "flame_param.npz").exists():
        modify_flame_params(str(src / "flame_param.npz"), str(dst / "flame_param.npz"), lefort_offset, bsso_offset,
                            deformation_map=deformation_map)
    if (src / "points3d.ply").exists():
        shutil.copy2(src / "points3d.ply", dst / "points3d.ply")
    if (src / "canonical_flame_param.npz").exists():       # selects the dynamic (FLAME-rigged) loader
        shutil.copy2(src / "canonical_flame_param.npz", dst / "canonical_flame_param.npz")
        print("[render_surgery] Copied canonical_flame_param.npz")

    for split in _SPLIT_FILES:
        if not (src / split).exists():
            continue
        with open(src / split, "r") as fh:
            transforms = json.load(fh)
        for frame in transforms.get("frames", []):
            candidate = f"flame_param/{frame.get('timestep_index', 0):05d}.npz"
            if (dst / candidate).exists():
                frame["flame_param_path"] = candidate
        with open(dst / split, "w") as fh:
            json.dump(transforms, fh, indent=2)

    print(f"[render_surgery] Modified dataset at: {temp_dir}")
    print("[render_surgery] Contents of temp_dir:")
    for item in os.listdir(temp_dir):
        full = os.path.join(temp_dir, item)
        print(f"  {item}/ ({len(os.listdir(full))} files)" if os.path.isdir(full) else f"  {item}")
    train_json = dst / "transforms_train.json"
    if train_json.exists():
        with open(train_json) as fh:
            t = json.load(fh)
        print(f"[render_surgery] transforms_train.json: {len(t.get('frames', []))} frames")
        if t.get("frames"):
            f0 = t["frames"][0]
            print(f"  First frame: timestep_index={f0.get('timestep_index')}, flame_param_path={f0.get('flame_param_path')}")
    return temp_dir


def _iteration_of(dirname: str) -> int:
    try:
        return int(dirname.split("_")[-1])
    except (ValueError, IndexError):
        return 0


def _count_png(folder: str) -> int:
    return sum(1 for f in os.listdir(folder) if f.endswith(".png"))


def build_render_command(model_path: str, data_dir: str, iteration: int, best_iteration: int | None) -> list[str]:
    """The engine argv of reference :289-301."""
    cmd = [sys.executable, str(RENDER_SCRIPT), "--source_path", os.path.abspath(data_dir),
           "--model_path", os.path.abspath(model_path), "--bind_to_mesh", "--skip_val", "--skip_test"]
    if iteration > 0:
        cmd += ["--iteration", str(iteration)]
    elif best_iteration:
        cmd += ["--iteration", str(best_iteration)]
    return cmd


def render_with_gaussians(model_path: str, data_dir: str, iteration: int = -1, clear_old_renders: bool = True) -> str:
    """Run the engine's render.py; return `<model>/train/ours_<iter>/renders` (reference :245-362)."""
    if not RENDER_SCRIPT.exists():
        raise FileNotFoundError(f"GaussianAvatars render.py not found at: {RENDER_SCRIPT}")

    train_dir = os.path.join(model_path, "train")
    if clear_old_renders and os.path.isdir(train_dir):          # never stitch stale frames
        for d in os.listdir(train_dir):
            stale = os.path.join(train_dir, d, "renders")
            if os.path.isdir(stale):
                print(f"[render_surgery] Clearing old renders: {stale}")
                shutil.rmtree(stale)

    best_iteration = None
    pc_dir = os.path.join(model_path, "point_cloud")
    if os.path.isdir(pc_dir):
        found = []
        for d in os.listdir(pc_dir):
            if d.startswith("iteration_"):
                try:
                    found.append(int(d.split("_")[1]))
                except (ValueError, IndexError):
                    pass
        if found:
            best_iteration = max(found)
            print(f"[render_surgery] Available iterations: {sorted(found)}")
            print(f"[render_surgery] Using iteration: {best_iteration}")

    cmd = build_render_command(model_path, data_dir, iteration, best_iteration)
    print("[render_surgery] Rendering with GaussianAvatars...")
    print(f"  Command: {' '.join(cmd)}")
    env = os.environ.copy()
    env["PYTHONPATH"] = str(REPO_DIR) + os.pathsep + env.get("PYTHONPATH", "")
    result = subprocess.run(cmd, cwd=str(REPO_DIR), env=env, capture_output=True, text=True)
    if result.returncode != 0:
        print(f"[render_surgery] Render stderr: {result.stderr[-2000:]}")
        print(f"[render_surgery] Render stdout: {result.stdout[-2000:]}")
        raise RuntimeError(f"Rendering failed:\n{result.stderr[-2000:]}")

    renders_dir = None
    if os.path.isdir(train_dir):
        if iteration > 0:
            wanted = os.path.join(train_dir, f"ours_{iteration}", "renders")
            if os.path.isdir(wanted):
                print(f"[render_surgery] Found renders at ours_{iteration}: {_count_png(wanted)} frames")
                print(f"[render_surgery] Frames rendered to: {wanted}")
                return wanted
        for d in sorted(os.listdir(train_dir), key=_iteration_of, reverse=True):
            candidate = os.path.join(train_dir, d, "renders")
            if os.path.isdir(candidate):
                print(f"[render_surgery] Found renders at {d}: {_count_png(candidate)} frames")
                renders_dir = candidate
                break
    if renders_dir is None:
        raise FileNotFoundError("No rendered frames found after GaussianAvatars rendering.")
    print(f"[render_surgery] Frames rendered to: {renders_dir}")
    return renders_dir


def select_deterministic_indices(n_frames: int, max_frames: int) -> list[int]:
    """Evenly spaced, de-duplicated: round(i*(n-1)/(k-1)), k = clamp(max_frames, 1, n) (reference :387-394)."""
    k = max(1, min(max_frames, n_frames))
    if k == 1:
        return [0]
    return sorted({int(round(i * (n_frames - 1) / (k - 1))) for i in range(k)})


def export_deterministic_frames(frames_dir: str, output_dir: str, index_file: str | None = None, max_frames: int = 24) -> str:
    """Copy a reproducible subset as idx_%05d.png + deterministic_indices_manifest.json (reference :365-409)."""
    os.makedirs(output_dir, exist_ok=True)
    frames = sorted(f for f in os.listdir(frames_dir) if f.endswith(".png"))
    if not frames:
        raise FileNotFoundError(f"No PNG frames in {frames_dir}")
    if index_file:
        with open(index_file, "r", encoding="utf-8") as fh:
            payload = json.load(fh)
        indices = payload.get("indices", payload)
        if not isinstance(indices, list) or not all(isinstance(i, int) for i in indices):
            raise ValueError("index_file must contain a JSON list of frame indices or {'indices': [...]} ")
        selected = [i for i in indices if 0 <= i < len(frames)]
    else:
        selected = select_deterministic_indices(len(frames), max_frames)

    manifest = {"source_frames_dir": frames_dir, "selected_indices": selected, "exports": []}
    for i in selected:
        exported = f"idx_{i:05d}.png"
        shutil.copy2(os.path.join(frames_dir, frames[i]), os.path.join(output_dir, exported))
        manifest["exports"].append({"index": i, "source": frames[i], "exported": exported})
    with open(os.path.join(output_dir, "deterministic_indices_manifest.json"), "w", encoding="utf-8") as fh:
        json.dump(manifest, fh, indent=2)
    print(f"[render_surgery] Deterministic frame export written to: {output_dir}")
    return output_dir


def stitch_video(frames_dir: str, output_path: str, fps: int = 30):
    """PNG frames -> H.264 MP4 with ffmpeg: libx264, yuv420p, preset medium, crf 18 (reference :412-449)."""
    ffmpeg_bin = _get_ffmpeg_path()
    parent = os.path.dirname(output_path)
    if parent:
        os.makedirs(parent, exist_ok=True)
    frames = sorted(f for f in os.listdir(frames_dir) if f.endswith(".png"))
    if not frames:
        raise FileNotFoundError(f"No PNG frames in {frames_dir}")
    staging = tempfile.mkdtemp(prefix="stitch_")
    for i, name in enumerate(frames):
        shutil.copy2(os.path.join(frames_dir, name), os.path.join(staging, f"frame_{i:05d}.png"))
    cmd = [ffmpeg_bin, "-y", "-framerate", str(fps), "-i", os.path.join(staging, "frame_%05d.png"),
           "-c:v", "libx264", "-pix_fmt", "yuv420p", "-preset", "medium", "-crf", "18", output_path]
    result = subprocess.run(cmd, capture_output=True, text=True)
    shutil.rmtree(staging, ignore_errors=True)
    if result.returncode != 0:
        raise RuntimeError(f"ffmpeg failed:\n{result.stderr}")
    print(f"[render_surgery] Video saved to {output_path}")


def main():
    p = argparse.ArgumentParser(description="Render post-surgical prediction video.")
    p.add_argument("--lefort_mm", type=float, required=True)
    p.add_argument("--bsso_mm", type=float, required=True)
    p.add_argument("--sensitivity", type=float, default=1.0)
    p.add_argument("--model_path", type=str, default="02_Visual_Engine/output/model")
    p.add_argument("--data_dir", type=str, default="02_Visual_Engine/data")
    p.add_argument("--output", type=str, default="final_prediction.mp4")
    p.add_argument("--fps", type=int, default=30)
    p.add_argument("--iteration", type=int, default=-1, help="Explicit model iteration to render.")
    p.add_argument("--rig_mode", type=str, default="flame_only", choices=("flame_only", "hybrid_full_head"),
                   help="Rendering rig mode. hybrid_full_head falls back to flame_only when asset is absent.")
    p.add_argument("--canonical_head_asset", type=str, default="", help="Path to canonical full-head asset used by hybrid_full_head mode.")
    p.add_argument("--deformation_map", type=str, default="", help="Optional JSON deformation map controlling region-aware scaling/axes.")
    p.add_argument("--export_frames_dir", type=str, default="", help="If set, exports deterministic frame subset for strict A/B evaluation.")
    p.add_argument("--deterministic_indices", type=str, default="", help="Optional JSON file with deterministic frame indices.")
    p.add_argument("--deterministic_max_frames", type=int, default=24, help="Max deterministic frames when indices are auto-generated.")
    args = p.parse_args()

    lefort_offset = compute_offset(args.lefort_mm, args.sensitivity)
    bsso_offset = compute_offset(args.bsso_mm, args.sensitivity)
    mode, reason = choose_rig_mode(args.rig_mode, args.canonical_head_asset)
    deformation_map = load_deformation_map(args.deformation_map if mode == "hybrid_full_head" else None)
    print(f"[render_surgery] Le Fort: {args.lefort_mm} mm -> offset {lefort_offset:.6f}")
    print(f"[render_surgery] BSSO:    {args.bsso_mm} mm -> offset {bsso_offset:.6f}")
    print(f"[render_surgery] Rig mode: {mode} ({reason})")
    if args.iteration > 0:
        print(f"[render_surgery] Pinned iteration: {args.iteration}")

    modified_dir = create_modified_dataset(args.data_dir, lefort_offset, bsso_offset, deformation_map=deformation_map)
    try:
        frames_dir = render_with_gaussians(args.model_path, modified_dir, iteration=args.iteration)
        if args.export_frames_dir:
            export_deterministic_frames(frames_dir=frames_dir, output_dir=args.export_frames_dir,
                                        index_file=args.deterministic_indices or None, max_frames=args.deterministic_max_frames)
        stitch_video(frames_dir, args.output, fps=args.fps)
    finally:
        shutil.rmtree(modified_dir, ignore_errors=True)     # the temporary dataset is ours to delete
    print("[render_surgery] Done.")


if __name__ == "__main__":
    main()

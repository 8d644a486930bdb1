"""train_ghost -- drop-in for `02_Visual_Engine/train_ghost.py` of the reference.

Same call surface (function names, arguments, defaults, messages, exceptions, CLI flags; reference
`train_ghost.py:31-301`), same engine argv (`:227-240`), same manifest (`:159-187`), same process
boundary: `train()` launches `<ENGINE_DIR>/train.py` as a child process and raises RuntimeError on
a non-zero exit code (`:262-276`).  The difference is what ENGINE_DIR is: not the un-vendored CUDA
checkout (`gaussian_avatars_repo`, `.gitignore:27`) but this package's `engine/` directory, whose
`train.py` runs the MI355X HIP kernels.  Set OMFS_ENGINE_DIR to point somewhere else.

    python -m omfs_4d_video_gen_amd.train_ghost --data_dir DATA --output_dir MODEL --iterations 30000
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import subprocess
import sys
from datetime import datetime, timezone
from pathlib import Path

# The reference calls this REPO_DIR / TRAIN_SCRIPT (train_ghost.py:27-28); names kept.
REPO_DIR = Path(os.environ.get("OMFS_ENGINE_DIR", Path(__file__).resolve().parent / "engine"))
TRAIN_SCRIPT = REPO_DIR / "train.py"

_REQUIRED_FILES = ("transforms_train.json", "transforms_test.json", "flame_param.npz")
_FINGERPRINT_FILES = ("transforms_train.json", "transforms_test.json", "transforms_val.json", "flame_param.npz",
                      "canonical_flame_param.npz")
MIN_TRAIN_FRAMES = 50


def validate_setup():
    """The engine directory and its train.py must exist (reference :31-43)."""
    if not REPO_DIR.exists():
        raise FileNotFoundError(
            f"GaussianAvatars repo not found at: {REPO_DIR}\n"
            "Clone it with:\n"
            "  git clone https://github.com/ShenhanQian/GaussianAvatars.git gaussian_avatars_repo")
    if not TRAIN_SCRIPT.exists():
        raise FileNotFoundError(f"train.py not found at: {TRAIN_SCRIPT}\nThe GaussianAvatars repo may be incomplete.")


def validate_data(data_dir: str):
    """Dataset must hold the three required files and at least one PNG (reference :46-65)."""
    for name in _REQUIRED_FILES:
        candidate = os.path.join(data_dir, name)
        if not os.path.exists(candidate):
            raise FileNotFoundError(f"Missing: {candidate}\nRun preprocess_video.py first to prepare the dataset.")
    images_dir = os.path.join(data_dir, "images")
    if not os.path.isdir(images_dir):
        raise FileNotFoundError(f"Images directory not found: {images_dir}\nRun preprocess_video.py first.")
    n_images = sum(1 for f in os.listdir(images_dir) if f.endswith(".png"))
    if n_images == 0:
        raise FileNotFoundError("No PNG frames found in images directory.")
    print(f"[train_ghost] Dataset validated: {n_images} frames")


def _sha256_file(path: Path) -> str:
    digest = hashlib.sha256()
    with open(path, "rb") as fh:
        for block in iter(lambda: fh.read(1 << 20), b""):
            digest.update(block)
    return digest.hexdigest()


def build_dataset_fingerprint(data_dir: str) -> dict:
    """sha256 per key file + sha256 of the sorted JSON of those hashes (reference :79-99)."""
    root = Path(data_dir)
    hashes = {rel: _sha256_file(root / rel) for rel in _FINGERPRINT_FILES if (root / rel).exists()}
    combined = hashlib.sha256(json.dumps(hashes, sort_keys=True).encode("utf-8")).hexdigest()
    return {"files": hashes, "dataset_hash": combined}


def run_quality_gates(data_dir: str):
    """Fail fast on thin, gappy or badly masked datasets (reference :102-138)."""
    root = Path(data_dir)
    with open(root / "transforms_train.json", "r", encoding="utf-8") as fh:
        frames = json.load(fh).get("frames", [])
    n = len(frames)
    if n < MIN_TRAIN_FRAMES:
        raise RuntimeError(f"Quality gate failed: only {n} training frames; need at least {MIN_TRAIN_FRAMES}.")
    steps = [int(fr.get("timestep_index", i)) for i, fr in enumerate(frames)]
    gaps = sum(1 for prev, cur in zip(steps, steps[1:]) if cur - prev > 1)
    if gaps > max(10, n // 10):
        raise RuntimeError(f"Quality gate failed: too many timeline gaps in train split ({gaps}).")
    masks = root / "fg_masks"
    if masks.exists():
        n_masks = sum(1 for p in masks.iterdir() if p.suffix.lower() == ".png")
        if n_masks < n // 2:
            raise RuntimeError(f"Quality gate failed: only {n_masks} fg masks for {n} train frames.")
    print(f"[train_ghost] Quality gates passed: frames={n}, timeline_gaps={gaps}")


def _collect_checkpoint_lineage(output_dir: str):
    out = Path(output_dir)
    if not out.exists():
        return []
    lineage = []
    for ckpt in sorted(out.glob("chkpnt*.pth")):
        st = ckpt.stat()
        lineage.append({"name": ckpt.name, "size_bytes": st.st_size,
                        "modified_utc": datetime.fromtimestamp(st.st_mtime, tz=timezone.utc).isoformat()})
    return lineage


def write_experiment_manifest(data_dir: str, output_dir: str, iterations: int, resolution: int, cmd: list[str], extra: dict):
    """<output_dir>/experiment_manifests/<UTC>.json (reference :159-187)."""
    folder = Path(output_dir) / "experiment_manifests"
    folder.mkdir(parents=True, exist_ok=True)
    stamp = datetime.now(timezone.utc)
    manifest_path = folder / f"{stamp.strftime('%Y%m%dT%H%M%SZ')}.json"
    payload = {
        "created_utc": datetime.now(timezone.utc).isoformat(),
        "data_dir": str(Path(data_dir).resolve()),
        "output_dir": str(Path(output_dir).resolve()),
        "iterations": iterations,
        "resolution": resolution,
        "command": cmd,
        "dataset_fingerprint": build_dataset_fingerprint(data_dir),
        "checkpoint_lineage": _collect_checkpoint_lineage(output_dir),
        "extra": extra,
    }
    with open(manifest_path, "w", encoding="utf-8") as fh:
        json.dump(payload, fh, indent=2)
    print(f"[train_ghost] Wrote experiment manifest: {manifest_path}")
    return manifest_path


def save_iterations_for(iterations: int) -> list[int]:
    """Final iteration always; midpoint from 5000; quarter from 10000 (reference :217-221)."""
    marks = [iterations]
    if iterations >= 5000:
        marks.insert(0, iterations // 2)
    if iterations >= 10000:
        marks.insert(0, iterations // 4)
    return marks


def build_train_command(data_dir: str, output_dir: str, iterations: int, resolution: int, has_masks: bool) -> list[str]:
    """The engine argv of reference :227-240."""
    marks = [str(i) for i in save_iterations_for(iterations)]
    cmd = [sys.executable, str(TRAIN_SCRIPT),
           "--source_path", os.path.abspath(data_dir),
           "--model_path", os.path.abspath(output_dir),
           "--bind_to_mesh",
           "--iterations", str(iterations),
           "--resolution", str(resolution),
           "--save_iterations", *marks,
           "--checkpoint_iterations", *marks]
    if has_masks:
        cmd.append("--white_background")
    return cmd


def train(data_dir: str, output_dir: str, iterations: int = 30000, resolution: int = -1):
    """Validate, gate, write the manifest and launch the engine's train.py (reference :190-278)."""
    validate_setup()
    validate_data(data_dir)
    run_quality_gates(data_dir)
    os.makedirs(output_dir, exist_ok=True)

    masks_dir = os.path.join(data_dir, "fg_masks")
    has_masks = os.path.isdir(masks_dir) and len(os.listdir(masks_dir)) > 0
    cmd = build_train_command(data_dir, output_dir, iterations, resolution, has_masks)
    if has_masks:
        print("[train_ghost] Using white background (fg_masks found)")
    else:
        print("[train_ghost] Using original background (no fg_masks)")

    write_experiment_manifest(data_dir=data_dir, output_dir=output_dir, iterations=iterations, resolution=resolution,
                              cmd=cmd, extra={"has_masks": has_masks})

    print("[train_ghost] Starting training:")
    print(f"  Data:       {data_dir}")
    print(f"  Output:     {output_dir}")
    print(f"  Iterations: {iterations}")
    print(f"  Command:    {' '.join(cmd)}")
    print()

    env = os.environ.copy()
    env["PYTHONPATH"] = str(REPO_DIR) + os.pathsep + env.get("PYTHONPATH", "")
    result = subprocess.run(cmd, cwd=str(REPO_DIR), env=env, text=True, capture_output=False)
    if result.returncode != 0:
        raise RuntimeError(f"Training failed with exit code {result.returncode}.")
    print(f"\n[train_ghost] Training complete! Model saved to: {output_dir}")


def main():
    parser = argparse.ArgumentParser(description="Train GaussianAvatars model.")
    parser.add_argument("--data_dir", type=str, default="02_Visual_Engine/data", help="Path to preprocessed dataset.")
    parser.add_argument("--output_dir", type=str, default="02_Visual_Engine/output/model", help="Path to save trained model.")
    parser.add_argument("--iterations", type=int, default=5000,
                        help="Training iterations (default 5000 for quick test, 30000 for good quality, 600000 for full).")
    parser.add_argument("--resolution", type=int, default=-1, help="Training resolution. -1 = native resolution (recommended).")
    args = parser.parse_args()
    train(args.data_dir, args.output_dir, args.iterations, args.resolution)


if __name__ == "__main__":
    main()

"""List capture sequences into sequence_manifest.json (reference `head_recon/ingest_sequences.py:18-44`)."""
from __future__ import annotations

import argparse
import json
from pathlib import Path

_IMAGE_SUFFIXES = (".jpg", ".png")


def _count_frames(images_dir: Path) -> int:
    return sum(1 for p in images_dir.iterdir() if p.suffix.lower() in _IMAGE_SUFFIXES) if images_dir.exists() else 0


def ingest_sequences(capture_root: Path, output_dir: Path) -> Path:
    output_dir.mkdir(parents=True, exist_ok=True)
    found = []
    for seq in sorted(p for p in capture_root.iterdir() if p.is_dir()):
        transforms, images = seq / "transforms_train.json", seq / "images"
        if transforms.exists() or images.exists():
            found.append({"name": seq.name, "path": str(seq.resolve()),
                          "transforms_train": str(transforms.resolve()) if transforms.exists() else "",
                          "image_count": _count_frames(images)})
    out_path = output_dir / "sequence_manifest.json"
    with open(out_path, "w", encoding="utf-8") as f:
        json.dump({"capture_root": str(capture_root.resolve()), "sequence_count": len(found), "sequences": found}, f, indent=2)
    print(f"[head_recon] Wrote sequence manifest: {out_path}")
    return out_path


def main():
    ap = argparse.ArgumentParser(description="Ingest multi-sequence captures.")
    ap.add_argument("--capture_root", required=True, type=Path)
    ap.add_argument("--output_dir", type=Path, default=Path("02_Visual_Engine/output/head_recon"))
    a = ap.parse_args()
    ingest_sequences(a.capture_root, a.output_dir)


if __name__ == "__main__":
    main()

"""canonical_head_asset.npz scaffold + manifest (reference `head_recon/build_canonical_head.py:14-44`);
its existence is what `render_surgery.choose_rig_mode` checks for hybrid_full_head."""
from __future__ import annotations

import argparse
import json
from pathlib import Path

import numpy as np

_NOTE = "Scaffold asset for hybrid_full_head mode. Replace with dense head geometry/field in future iterations."


def build_canonical_head(registration_path: Path, output_dir: Path) -> tuple[Path, Path]:
    with open(registration_path, "r", encoding="utf-8") as f:
        reg = json.load(f)
    canonical = reg.get("canonical_sequence", "unknown")
    output_dir.mkdir(parents=True, exist_ok=True)
    asset = output_dir / "canonical_head_asset.npz"
    np.savez(asset, version=np.array([1], dtype=np.int32), canonical_sequence=np.array([canonical]),
             registration_count=np.array([len(reg.get("registrations", []))], dtype=np.int32))
    manifest_path = output_dir / "canonical_head_asset_manifest.json"
    with open(manifest_path, "w", encoding="utf-8") as f:
        json.dump({"canonical_sequence": canonical, "registration_source": str(registration_path.resolve()),
                   "asset_path": str(asset.resolve()), "notes": _NOTE}, f, indent=2)
    print(f"[head_recon] Wrote canonical head scaffold: {asset}")
    print(f"[head_recon] Wrote canonical asset manifest: {manifest_path}")
    return asset, manifest_path


def main():
    ap = argparse.ArgumentParser(description="Build canonical full-head scaffold asset.")
    ap.add_argument("--registration", required=True, type=Path)
    ap.add_argument("--output_dir", type=Path, default=Path("02_Visual_Engine/output/head_recon"))
    a = ap.parse_args()
    build_canonical_head(a.registration, a.output_dir)


if __name__ == "__main__":
    main()

"""Registration table: first sequence is canonical, all transforms identity
(reference `head_recon/register_sequences.py:12-48`; a placeholder policy there too)."""
from __future__ import annotations

import argparse
import json
from pathlib import Path

_IDENTITY = [[1.0 if r == c else 0.0 for c in range(4)] for r in range(4)]


def register_sequences(manifest_path: Path, output_dir: Path) -> Path:
    with open(manifest_path, "r", encoding="utf-8") as f:
        sequences = json.load(f).get("sequences", [])
    if not sequences:
        raise RuntimeError("No sequences found in manifest.")
    canonical = sequences[0]["name"]
    rows = [{"sequence": s["name"], "to_canonical_transform": [list(r) for r in _IDENTITY],
             "confidence": 1.0 if s["name"] == canonical else 0.7} for s in sequences]
    output_dir.mkdir(parents=True, exist_ok=True)
    out_path = output_dir / "registration.json"
    with open(out_path, "w", encoding="utf-8") as f:
        json.dump({"canonical_sequence": canonical, "manifest_path": str(manifest_path.resolve()), "registrations": rows}, f, indent=2)
    print(f"[head_recon] Wrote registration: {out_path}")
    return out_path


def main():
    ap = argparse.ArgumentParser(description="Register capture sequences to canonical frame.")
    ap.add_argument("--manifest", required=True, type=Path)
    ap.add_argument("--output_dir", type=Path, default=Path("02_Visual_Engine/output/head_recon"))
    a = ap.parse_args()
    register_sequences(a.manifest, a.output_dir)


if __name__ == "__main__":
    main()

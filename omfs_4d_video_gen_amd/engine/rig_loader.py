"""Where the engine finds its FLAME rig.

The reference expects the licensed FLAME 2023 pickle at
`gaussian_avatars_repo/flame_model/assets/flame/flame2023.pkl` (`02_Visual_Engine/flame_fitter.py:37-39`,
git-ignored at `.gitignore:28-29`).  The engine looks, in order, at $OMFS_FLAME_PKL, that same
relative path under the engine directory, and -- only when OMFS_SYNTHETIC_RIG=1 -- falls back to the
seeded synthetic rig used by the tests and the benchmark.  Anything else is a FileNotFoundError.
"""
from __future__ import annotations

import os
from pathlib import Path

from .flame_rig import FlameRig

ENGINE_DIR = Path(__file__).resolve().parent
FLAME_MODEL_PATH = ENGINE_DIR / "flame_model" / "assets" / "flame" / "flame2023.pkl"


def load_rig(n_verts_hint: int | None = None) -> FlameRig:
    explicit = os.environ.get("OMFS_FLAME_PKL")
    for cand in ([Path(explicit)] if explicit else []) + [FLAME_MODEL_PATH]:
        if cand.exists():
            # teeth: lip rings from $OMFS_FLAME_LIP_RINGS (npz with `upper`, `lower`: 15 vertex ids each), else from
            # FLAME_masks.pkl next to the model (flame_rig.lip_rings_from_masks), else the rig keeps the pickle's 5023 vertices
            rings = os.environ.get("OMFS_FLAME_LIP_RINGS")
            lip = None
            if rings:
                import numpy as np
                z = np.load(rings)
                lip = (z["upper"], z["lower"])
            return FlameRig.from_pickle(str(cand), lip_rings=lip)
    if os.environ.get("OMFS_SYNTHETIC_RIG") == "1":
        from . import synthetic
        return FlameRig.from_synthetic(synthetic.make_rig(int(os.environ.get("OMFS_SYNTHETIC_RIG_SEED", "0"))))
    raise FileNotFoundError(
        f"FLAME model not found at: {FLAME_MODEL_PATH}\n"
        "Copy flame2023.pkl there, point OMFS_FLAME_PKL at it, or set OMFS_SYNTHETIC_RIG=1 to use the synthetic test rig.")

"""Shared builders for tests: same seeded inputs for the HIP path and the oracle."""
import numpy as np
import torch

from omfs_4d_video_gen_amd.engine import synthetic
from omfs_4d_video_gen_amd.engine.flame_rig import FlameRig, pose_rotmats


def oracle_rig(rig):
    """synthetic rig -> dict of torch tensors for oracle.torch_splat."""
    return {"v_template": torch.from_numpy(rig.v_template), "shapedirs": torch.from_numpy(rig.shapedirs),
            "posedirs": torch.from_numpy(rig.posedirs), "J_regressor": torch.from_numpy(rig.J_regressor),
            "weights": torch.from_numpy(rig.weights), "faces": torch.from_numpy(rig.faces.astype(np.int64))}


def oracle_frame(seq, t):
    rm = pose_rotmats(seq)
    return {"shape": torch.from_numpy(seq["shape"]), "expr": torch.from_numpy(seq["expr"][t]),
            "rotmats": torch.from_numpy(rm[t]), "translation": torch.from_numpy(seq["translation"][t]),
            "static_offset": torch.from_numpy(seq["static_offset"][0]),
            "dynamic_offset": torch.from_numpy(seq["dynamic_offset"][t])}


def oracle_gaussians(g, requires_grad=False):
    d = {k: torch.from_numpy(np.asarray(v)).clone() for k, v in g.items()}
    if requires_grad:
        for k in ("xyz", "log_scale", "rot", "opacity", "sh"):
            d[k].requires_grad_(True)
    return d

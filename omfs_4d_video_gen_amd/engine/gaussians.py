"""Mesh-bound Gaussian cloud as one planar SoA in HBM: params[59][n_pad] fp32 + binding[n] int32.

Plane order = include/omfs_splat.h OMFS_P_*: xyz(3) log_scale(3) rot wxyz(4) opacity(1)
SH k,c at 11+3k+c (48).  One contiguous buffer per role (params, grads, Adam m, Adam v), so the
optimiser is a single elementwise pass and the data-parallel gradient exchange is ONE RCCL
all-reduce over one buffer.
"""
from __future__ import annotations

import numpy as np
import torch

NPLANES = 59
P_XYZ, P_SCALE, P_ROT, P_OPACITY, P_SH = 0, 3, 6, 10, 11


def pack_params(g: dict) -> np.ndarray:
    """dict of (N,...) arrays (synthetic.make_gaussians schema) -> [59][N] float32."""
    n = g["xyz"].shape[0]
    out = np.empty((NPLANES, n), np.float32)
    out[P_XYZ:P_XYZ + 3] = np.asarray(g["xyz"], np.float32).T
    out[P_SCALE:P_SCALE + 3] = np.asarray(g["log_scale"], np.float32).T
    out[P_ROT:P_ROT + 4] = np.asarray(g["rot"], np.float32).T
    out[P_OPACITY] = np.asarray(g["opacity"], np.float32)
    out[P_SH:] = np.asarray(g["sh"], np.float32).reshape(n, 48).T     # (N,16,3) -> plane 3k+c
    return out


def unpack_params(p: np.ndarray) -> dict:
    n = p.shape[1]
    return {"xyz": p[P_XYZ:P_XYZ + 3].T.copy(), "log_scale": p[P_SCALE:P_SCALE + 3].T.copy(),
            "rot": p[P_ROT:P_ROT + 4].T.copy(), "opacity": p[P_OPACITY].copy(),
            "sh": p[P_SH:].T.reshape(n, 16, 3).copy()}


class GaussianModel:
    def __init__(self, g: dict, device="cuda"):
        self.device = torch.device(device)
        self.n = int(g["xyz"].shape[0])
        self.n_pad = (self.n + 255) // 256 * 256
        host = np.zeros((NPLANES, self.n_pad), np.float32)
        host[:, :self.n] = pack_params(g)
        host[P_ROT, self.n:] = 1.0            # keep padded quaternions normalisable
        self.params = torch.from_numpy(host).to(self.device)
        self.binding = torch.from_numpy(np.asarray(g["binding"], np.int32)).to(self.device)

    def to_dict(self) -> dict:
        d = unpack_params(self.params[:, :self.n].cpu().numpy())
        d["binding"] = self.binding.cpu().numpy()
        return d

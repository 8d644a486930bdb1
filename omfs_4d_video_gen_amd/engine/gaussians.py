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


def coherent_order(binding: np.ndarray, face_centres: np.ndarray) -> np.ndarray:
    """Permutation that lays the Gaussians out along a Morton curve over their parent triangles' rest-pose centres (stable
    inside a triangle).  Datasets bind Gaussian i to triangle i mod F, so 512 consecutive Gaussians are spread over the
    whole head and a binning workgroup touches thousands of tiles; in this order they cover a patch of the surface -- a few
    tiles -- so the per-(workgroup, tile) aggregation of the binning kernels really aggregates, and the gathers of triangle
    records and projected splats hit neighbouring lines.  The cloud is a set: the order carries no meaning."""
    c = np.asarray(face_centres, np.float64)
    lo, span = c.min(0), np.maximum(c.max(0) - c.min(0), 1e-12)
    q = np.minimum(((c - lo) / span * 1023.0).astype(np.uint64), 1023)

    def spread(v):          # 10 bits -> every third bit
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        return (v | (v << 2)) & 0x09249249
    code = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
    return np.argsort(code[np.asarray(binding, np.int64)], kind="stable")


class GaussianModel:
    def __init__(self, g: dict, device="cuda", order: np.ndarray | None = None):
        """order (optional permutation): the cloud is stored as g[order]; `to_dict()` hands it back in the caller's order."""
        self.device = torch.device(device)
        self.n = int(g["xyz"].shape[0])
        self.n_pad = (self.n + 255) // 256 * 256
        self.order = None if order is None else np.asarray(order, np.int64)
        packed, binding = pack_params(g), np.asarray(g["binding"], np.int32)
        if self.order is not None:
            packed, binding = packed[:, self.order], binding[self.order]
        host = np.zeros((NPLANES, self.n_pad), np.float32)
        host[:, :self.n] = packed
        host[P_ROT, self.n:] = 1.0            # keep padded quaternions normalisable
        self.params = torch.from_numpy(host).to(self.device)
        self.binding = torch.from_numpy(np.ascontiguousarray(binding)).to(self.device)

    def caller_order(self, t: torch.Tensor) -> torch.Tensor:
        """Per-Gaussian columns [..., >= n] of a buffer that lies in storage order (parameters, Adam moments, statistics),
        as a CPU tensor [..., n] in the order the cloud was given in."""
        c = t[..., :self.n].detach().cpu()
        if self.order is None or self.order.shape[0] != self.n:
            return c
        inv = np.empty_like(self.order)
        inv[self.order] = np.arange(self.n)
        return c[..., torch.from_numpy(inv)]

    def to_dict(self, storage_order: bool = False) -> dict:
        """The cloud in the order it was given (storage_order: as it lies in HBM)."""
        p, b = self.params[:, :self.n].cpu().numpy(), self.binding.cpu().numpy()
        if self.order is not None and not storage_order and self.order.shape[0] == self.n:
            inv = np.empty_like(self.order)
            inv[self.order] = np.arange(self.n)
            p, b = p[:, inv], b[inv]
        d = unpack_params(p)
        d["binding"] = b
        return d

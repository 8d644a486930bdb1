"""FLAME-parameter fine-tuning: the per-timestep expression, joint poses and translation are optimised together
with the Gaussians, as upstream GaussianAvatars does by default for `--bind_to_mesh` (the
reference only launches that trainer: `02_Visual_Engine/train_ghost.py:227-240`).

Gradient path: composite_bwd -> project_bwd (dL/d triangle-frame record, `omfs_grad_buffers.dface`) ->
`omfs_face_frames_bwd` (dL/d posed vertices) -> `omfs_flame_skin_bwd` (dL/d blend-shaped vertices, dL/d joint
transforms, dL/d translation).  What remains is tiny (a 5-joint kinematic chain, the axis-angle map `rodrigues` of
`flame_fitter.py:122-152`, one [K]x[3V] basis product): it is evaluated here with torch ops on the device and
differentiated by autograd, which chains the three kernel outputs into the parameter gradients.
"""
from __future__ import annotations

import numpy as np
import torch

from .. import _lib as L
from .flame_rig import DeviceFlame, rodrigues

# upstream's learning rates for the FLAME parameter groups
FLAME_LR = {"expr": 1e-3, "pose": 1e-5, "translation": 1e-6}


class FlameFineTuner:
    def __init__(self, dflame: DeviceFlame, flame_params: dict, lr: dict | None = None):
        self.df = dflame
        dev = dflame.device
        T = dflame.n_frames

        def as2d(key, w):
            a = flame_params.get(key)
            a = np.zeros((T, w), np.float32) if a is None else np.asarray(a, np.float32).reshape(-1, w)
            return a
        eyes = as2d("eyes_pose", 6)
        pose = np.stack([as2d("rotation", 3), as2d("neck_pose", 3), as2d("jaw_pose", 3), eyes[:, :3], eyes[:, 3:]], 1)
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
        self.pose = up(pose).requires_grad_(True)                       # [T][5][3] axis-angle
        self.expr = up(dflame.h_expr).requires_grad_(True)              # [T][E]
        self.translation = up(dflame.h_translation).requires_grad_(True)
        self.source = {"pose": pose.copy(), "expr": dflame.h_expr.copy(), "translation": dflame.h_translation.copy()}
        lr = {**FLAME_LR, **(lr or {})}
        self.opt = torch.optim.Adam([{"params": [self.expr], "lr": lr["expr"]}, {"params": [self.pose], "lr": lr["pose"]},
                                     {"params": [self.translation], "lr": lr["translation"]}], eps=1e-15)
        V = dflame.rig.n_verts
        self.n_verts = V
        # dense basis [K][3V] (column v*3+c) for the blend-shape product and its transpose product
        self.basis = up(dflame.h_basis.transpose(0, 2, 1).reshape(dflame.h_basis.shape[0], 3 * V))
        self.v_static = up(dflame.h_v_static[:, :V].T.reshape(-1))
        self.j_static = up(dflame.h_j_static)                            # (5,3)
        self.j_expr = up(dflame.h_j_expr)                                # (15,E)
        self.eye = torch.eye(3, device=dev)
        F = dflame.rig.n_faces
        self.dface = torch.zeros(F, 16, device=dev)
        self.dverts = torch.zeros(dflame.v_pad, 4, device=dev)
        self.dv_shaped = torch.empty(V, 3, device=dev)
        self.sums = torch.zeros(64, device=dev)
        self._live = None

    # ---- forward: the small differentiable front; refreshes the rows the FLAME kernels read
    def begin(self, t: int):
        R = rodrigues(self.pose[t])                                      # (5,3,3)
        psi = self.expr[t]
        J = self.j_static + (self.j_expr @ psi).view(5, 3)
        Rw, tw = [R[0]], [J[0]]
        for j in range(1, 5):
            p = 0 if j == 1 else 1
            Rw.append(Rw[p] @ R[j])
            tw.append(Rw[p] @ (J[j] - J[p]) + tw[p])
        X = torch.cat([torch.cat([Rw[j].reshape(9), tw[j] - Rw[j] @ J[j]]) for j in range(5)])   # [60]
        coef = torch.cat([psi, (R[1:] - self.eye).reshape(36)])
        v_shaped = torch.addmv(self.v_static, self.basis.t(), coef)     # [3V], v*3+c
        with torch.no_grad():
            self.df.rotmats[t].copy_(R.reshape(45))
            self.df.expr[t].copy_(psi)
            self.df.translation[t].copy_(self.translation[t])
        self._live = (t, X, v_shaped)
        self.dface.zero_()

    # ---- backward: kernels for the vertex-sized work, autograd for the chain
    def backward(self, verts: torch.Tensor):
        """verts: [v_pad][4] posed vertices of the frame (DeviceFlame.face_frames); self.dface filled by project_bwd."""
        t, X, v_shaped = self._live
        lib, s = L.load(), L.stream_ptr()
        self.dverts.zero_()
        self.sums.zero_()
        L.check(lib.omfs_face_frames_bwd(L.ptr(verts), self.df.v_pad, L.ptr(self.df.faces), self.df.rig.n_faces,
                                         L.ptr(self.dface), L.ptr(self.dverts), s), "omfs_face_frames_bwd")
        Xd, vsd = X.detach().contiguous(), v_shaped.detach().contiguous()
        L.check(lib.omfs_flame_skin_bwd(self.df.c_rig, L.ptr(vsd), L.ptr(Xd), L.ptr(self.dverts), L.ptr(self.dv_shaped),
                                        L.ptr(self.sums), s), "omfs_flame_skin_bwd")
        surrogate = (v_shaped * self.dv_shaped.reshape(-1)).sum() + (X * self.sums[:60]).sum() + \
            (self.translation[t] * self.sums[60:63]).sum()
        surrogate.backward()
        self._live = None

    def grads(self):
        return [p.grad for p in (self.expr, self.pose, self.translation)]

    def step(self, grad_scale: float = 1.0):
        if grad_scale != 1.0:
            for p in (self.expr, self.pose, self.translation):
                if p.grad is not None:
                    p.grad.mul_(grad_scale)
        self.opt.step()
        self.opt.zero_grad(set_to_none=False)

    # ---- egress: the dataset schema of flame_fitter.py:431-441
    def to_flame_params(self, base: dict) -> dict:
        out = {k: np.array(v) for k, v in base.items()}
        pose = self.pose.detach().cpu().numpy()
        T = pose.shape[0]
        out["rotation"] = pose[:, 0].astype(np.float32)
        out["neck_pose"] = pose[:, 1].astype(np.float32)
        out["jaw_pose"] = pose[:, 2].astype(np.float32)
        out["eyes_pose"] = pose[:, 3:5].reshape(T, 6).astype(np.float32)
        e = np.array(base["expr"], np.float32).reshape(T, -1)
        e[:, :self.expr.shape[1]] = self.expr.detach().cpu().numpy()
        out["expr"] = e
        out["translation"] = self.translation.detach().cpu().numpy().astype(np.float32)
        return out

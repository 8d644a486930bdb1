"""FLAME-parameter fine-tuning: the per-timestep expression, joint poses and translation are optimised together
with the Gaussians, as upstream GaussianAvatars does by default for `--bind_to_mesh` (the
reference only launches that trainer: `02_Visual_Engine/train_ghost.py:227-240`).

Gradient path, all HIP (include/omfs_splat.h): composite_bwd -> project_bwd (dL/d triangle-frame record,
`omfs_grad_buffers.dface`) -> `omfs_face_frames_bwd` (dL/d posed vertices) -> `omfs_flame_skin_bwd` (dL/d blend-shaped
vertices, dL/d joint transforms, dL/d translation) -> `omfs_flame_param_bwd` (basis^T product, 5-joint kinematic chain,
axis-angle map of `flame_fitter.py:122-152`) -> `omfs_adam_flat`.  The parameters ARE the device arrays the FLAME
forward kernels read (`DeviceFlame.expr / .translation`, rotation matrices refreshed from the axis-angle poses by
the joints launch, `omfs_flame_joints_pose`), so nothing is copied per step.  Per step: three forward launches (joints +
Rodrigues, LBS, triangle frames), three backward launches (`dverts` and the gradient tensors are consumed = left zeroed
by their readers, the basis^T product and the serial front share a launch) and ONE Adam launch for the three parameter
tensors; no host math.  The trainer runs this chain on a second stream underneath the Gaussians' Adam pass.
"""

from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from .. import _lib as L
from .flame_rig import DeviceFlame

# upstream's learning rates for the FLAME parameter groups
FLAME_LR = {"expr": 1e-3, "pose": 1e-5, "translation": 1e-6}


class FlameFineTuner:
    def __init__(self, dflame: DeviceFlame, flame_params: dict, lr: dict | None = None,
                 beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-15):
        self.df = dflame
        dev = dflame.device
        T = dflame.n_frames

        def as2d(key, w):
            a = flame_params.get(key)
            return np.zeros((T, w), np.float32) if a is None else np.asarray(a, np.float32).reshape(-1, w)
        eyes = as2d("eyes_pose", 6)
        pose = np.stack([as2d("rotation", 3), as2d("neck_pose", 3), as2d("jaw_pose", 3), eyes[:, :3], eyes[:, 3:]], 1)
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
        self.pose = up(pose.reshape(T, 15))                 # [T][5*3] axis-angle: global, neck, jaw, eye-L, eye-R
        self.expr = dflame.expr                             # [T][E]   the arrays the forward kernels read
        self.translation = dflame.translation               # [T][3]
        self.params = {"expr": self.expr, "pose": self.pose, "translation": self.translation}
        # the three gradient tensors are views of ONE buffer: data-parallel ranks sum them with a single small all-reduce
        # (three latency-bound collectives per iteration otherwise)
        self.n_grad = sum(v.numel() for v in self.params.values())
        self._bind_grads(torch.zeros(self.n_grad, device=dev))
        self.m = {k: torch.zeros_like(v) for k, v in self.params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in self.params.items()}
        self.lr = {**FLAME_LR, **(lr or {})}
        self.betas, self.eps = (beta1, beta2), eps
        self.step_count = 0
        V = dflame.rig.n_verts
        self.n_coef = dflame.h_basis.shape[0]
        # dense basis [K][3V] (column 3v+c): blend-shape recompute per vertex and the transpose product
        self.basis = up(dflame.h_basis.transpose(0, 2, 1).reshape(self.n_coef, 3 * V))
        # ... and transposed, [3V][K] (row 3v+c): what the fused backward reads -- a wave's loads are contiguous in k and no
        # cross-lane reduction is needed (omfs_flame_skin_param_bwd)
        self.basis_t = up(dflame.h_basis.transpose(2, 1, 0).reshape(3 * V, self.n_coef))
        self.fused = os.environ.get("OMFS_FLAME_SPLIT", "0") != "1"
        # OMFS_DETERMINISTIC=1: the split launches (per-wave partial rows and block reductions in a fixed order: no float atomics)
        # behind a fixed-point omfs_face_frames_bwd_fx -- the FLAME gradients are then bit-reproducible from run to run
        self.deterministic = os.environ.get("OMFS_DETERMINISTIC", "0") == "1"
        self.dverts_fx = None
        if self.deterministic:
            self.fused = False
            self.dverts_fx = torch.zeros(dflame.v_pad, 4, dtype=torch.int64, device=dflame.device)
        dflame.keep_v_shaped = True
        dflame.pose = self.pose                              # the joints launch refreshes rotmats from these
        dflame._scratch.clear()
        self.dface = None                                    # [n_capacity][16] per-Gaussian frame-gradient records
        self._csr_key, self.face_start, self.face_gauss = None, None, None
        self.dverts = torch.zeros(dflame.v_pad, 4, device=dev)
        self.dv_shaped = torch.empty(V, 3, device=dev)
        lib = L.load()
        # scratch of the fused backward (accumulator copies + ticket, kept zero by the kernel); the split launches use the head of
        # the same buffers: one row of sums per wave, n_coef + 1 words of dcoef
        self.sums = torch.zeros(max(int(lib.omfs_flame_skin_rows(dflame.c_rig)), lib.omfs_flame_skin_param_scratch_floats(1) // 64), 64, device=dev)
        self.dcoef = torch.zeros(max(self.n_coef + 1, lib.omfs_flame_skin_param_scratch_floats(0)), device=dev)
        self._t = None
        self.refresh_rotmats()
        # host-side argument arrays of the one Adam launch (device pointers of the three tensors, sizes, learning rates)
        keys = ("expr", "pose", "translation")
        arr = lambda seq: (C.c_void_p * 3)(*[L.ptr(t) for t in seq])
        self._adam_args = (arr([self.params[k] for k in keys]), arr([self.grad[k] for k in keys]), arr([self.m[k] for k in keys]),
                           arr([self.v[k] for k in keys]), (C.c_int * 3)(*[self.params[k].numel() for k in keys]),
                           (C.c_float * 3)(*[float(self.lr[k]) for k in keys]))

    def _bind_grads(self, flat: torch.Tensor):
        self.grad_flat = flat
        self.grad, off = {}, 0
        for k, v in self.params.items():
            self.grad[k] = flat[off:off + v.numel()].view(v.shape)
            off += v.numel()

    def rebind_grads(self, flat: torch.Tensor):
        """Move the three gradient tensors into `flat` (zeroed, >= n_grad floats): the data-parallel trainer puts them in front
        of the Gaussians' gradient planes, in ONE allocation, so that one all-reduce sums both (a small collective of its own
        costs a launch, two stream hand-overs and its latency on the links every iteration)."""
        if flat.numel() < self.n_grad or flat.dtype != torch.float32:
            raise ValueError("gradient buffer too small")
        flat[:self.n_grad].copy_(self.grad_flat)
        self._bind_grads(flat[:self.n_grad])
        keys = ("expr", "pose", "translation")
        a = self._adam_args
        self._adam_args = (a[0], (C.c_void_p * 3)(*[L.ptr(self.grad[k]) for k in keys]), a[2], a[3], a[4], a[5])

    def refresh_rotmats(self):
        """All timesteps: axis-angle poses -> the rotation matrices the forward kernels read."""
        T = self.pose.shape[0]
        L.check(L.load().omfs_flame_rodrigues(L.ptr(self.pose), T * 5, L.ptr(self.df.rotmats), L.stream_ptr()), "omfs_flame_rodrigues")

    def bind(self, binding: torch.Tensor):
        """(Re)build the triangle -> Gaussians CSR and the record buffer when the cloud's topology changed."""
        key = (binding.data_ptr(), int(binding.shape[0]))
        if key == self._csr_key:
            return
        b = binding.to(torch.int64)
        order = torch.argsort(b, stable=True)
        counts = torch.bincount(b, minlength=self.df.rig.n_faces)
        self.face_start = torch.cat([torch.zeros(1, dtype=torch.int64, device=b.device), torch.cumsum(counts, 0)]).to(torch.int32)
        self.face_gauss = order.to(torch.int32).contiguous()
        if self.dface is None or self.dface.shape[0] < key[1]:
            self.dface = torch.empty(key[1], 16, device=binding.device)
        self._csr_key = key

    def begin(self, t: int, binding: torch.Tensor, all_timesteps: bool = False):
        """Before the FLAME forward of timestep t (the forward itself turns the current poses of the rows it shows into
        rotation matrices: `DeviceFlame.pose`)."""
        self.bind(binding)
        self._t = t

    def backward(self, verts: torch.Tensor, nb: int = 1, col: int = 0):
        """verts: [v_pad][4] posed vertices of the frame (column `col` of the last FLAME forward with batch nb);
        self.dface filled by project_bwd.  Leaves the gradient of timestep t in row t of the dense gradient tensors
        (all other rows zero)."""
        t = self._t
        df = self.df
        lib, s = L.load(), L.stream_ptr()
        joint_all, _, _, _, vs_all = df._buffers(nb)        # written by the FLAME forward kernels
        joint_xf, v_shaped = joint_all[col], vs_all[col]
        # dverts was left zeroed by the last omfs_flame_skin_bwd, the gradient rows by the last Adam launch
        if self.dverts_fx is not None:
            L.check(lib.omfs_face_frames_bwd_fx(L.ptr(verts), df.v_pad, L.ptr(df.faces), df.rig.n_faces, L.ptr(self.dface), L.ptr(self.face_start),
                                                L.ptr(self.face_gauss), L.ptr(self.dverts), L.ptr(self.dverts_fx), s), "omfs_face_frames_bwd_fx")
        else:
            L.check(lib.omfs_face_frames_bwd(L.ptr(verts), df.v_pad, L.ptr(df.faces), df.rig.n_faces, L.ptr(self.dface),
                                             L.ptr(self.face_start), L.ptr(self.face_gauss), L.ptr(self.dverts), s), "omfs_face_frames_bwd")
        if self.fused:     # skinning backward + basis^T product + serial front: one launch
            L.check(lib.omfs_flame_skin_param_bwd(df.c_rig, L.ptr(self.basis_t), self.n_coef, L.ptr(v_shaped), L.ptr(joint_xf),
                                                  L.ptr(self.dverts), L.ptr(self.expr[t]), L.ptr(self.pose[t]), L.ptr(self.dcoef),
                                                  L.ptr(self.sums), L.ptr(self.grad["expr"][t]), L.ptr(self.grad["pose"][t]),
                                                  L.ptr(self.grad["translation"][t]), s), "omfs_flame_skin_param_bwd")
            self._t = None
            return
        L.check(lib.omfs_flame_skin_bwd(df.c_rig, L.ptr(v_shaped), L.ptr(joint_xf), L.ptr(self.dverts), L.ptr(self.dv_shaped),
                                        L.ptr(self.sums), s), "omfs_flame_skin_bwd")
        L.check(lib.omfs_flame_param_bwd(df.c_rig, L.ptr(self.basis), self.n_coef, L.ptr(self.dv_shaped), L.ptr(self.expr[t]),
                                         L.ptr(self.pose[t]), L.ptr(self.sums), L.ptr(self.dcoef), L.ptr(self.grad["expr"][t]),
                                         L.ptr(self.grad["pose"][t]), L.ptr(self.grad["translation"][t]), s), "omfs_flame_param_bwd")
        self._t = None

    def grads(self):
        return [self.grad["expr"], self.grad["pose"], self.grad["translation"]]

    def step(self, grad_scale: float = 1.0, state_dev: int = 0):
        """Dense Adam over all timesteps (torch.optim.Adam semantics, as upstream).  state_dev: device pointer of an
        omfs_step_state whose flame_step / bias corrections replace the host-side step count (graph replay; the caller keeps
        `step_count` in step with it)."""
        if not state_dev:
            self.step_count += 1
        a = self._adam_args
        L.check(L.load().omfs_adam_flat_multi(3, a[0], a[1], a[2], a[3], a[4], a[5], self.betas[0], self.betas[1], self.eps,
                                              max(self.step_count, 1), float(grad_scale), state_dev, L.stream_ptr()), "omfs_adam_flat_multi")

    # ---- checkpoint / egress
    def state_dict(self) -> dict:
        torch.cuda.synchronize(self.df.device)      # the parameters are updated on the trainer's side stream
        cpu = lambda d: {k: v.detach().cpu() for k, v in d.items()}
        return {"params": cpu(self.params), "m": cpu(self.m), "v": cpu(self.v), "step": self.step_count}

    def load_state_dict(self, st: dict):
        for k in self.params:
            self.params[k].copy_(st["params"][k]); self.m[k].copy_(st["m"][k]); self.v[k].copy_(st["v"][k])
        self.step_count = int(st["step"])
        self.refresh_rotmats()

    def to_flame_params(self, base: dict) -> dict:
        """The tuned sequence in the dataset schema of flame_fitter.py:431-441."""
        torch.cuda.synchronize(self.df.device)      # the parameters are updated on the trainer's side stream
        out = {k: np.array(v) for k, v in base.items()}
        pose = self.pose.detach().cpu().numpy().reshape(-1, 5, 3)
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

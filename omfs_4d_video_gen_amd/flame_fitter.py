"""flame_fitter -- drop-in for `02_Visual_Engine/flame_fitter.py` of the reference: fit the simplified
FLAME landmark model to 2-D face landmarks and write `flame_param.npz` in the dataset schema
(`flame_fitter.py:5-12, 431-441`).

Same call surface: `MEDIAPIPE_TO_68`, `SimpleFLAME(flame_model_path, n_shape, n_expr)` with
`.forward(shape, expr, rotation, jaw, translation)`, `detect_landmarks_mediapipe`,
`estimate_head_pose_from_landmarks`, `fit_flame_to_landmarks`, `fit_video`, CLI flags.
`SimpleFLAME.forward` and its gradient run in HIP kernels (`csrc/simple_flame.hip`, C ABI
`omfs_simpleflame_fwd/bwd`) behind a `torch.autograd.Function`; the barycentric landmark mix is
folded into a 68-landmark basis on the host, so each iteration touches 68 points, not 5023 vertices.
The fit loop itself is one C-ABI call per iteration (`omfs_flame_fit_step`: forward, masked MSE of the
pseudo-perspective projection, closed-form gradients including the regularisers and the temporal smoothness,
Adam) -- no autograd.  There is no CPU path: tensors must live on the GPU.
"""
from __future__ import annotations

import argparse
import os
from pathlib import Path

import numpy as np
import torch

from . import _lib as L

REPO_DIR = Path(os.environ.get("OMFS_ENGINE_DIR", Path(__file__).resolve().parent / "engine"))
FLAME_MODEL_PATH = REPO_DIR / "flame_model" / "assets" / "flame" / "flame2023.pkl"
FLAME_LMK_PATH = REPO_DIR / "flame_model" / "assets" / "flame" / "landmark_embedding_with_eyes.npy"

# 68 standard face landmarks as MediaPipe FaceMesh indices (reference :45-66)
MEDIAPIPE_TO_68 = (
    [10, 338, 297, 332, 284, 251, 389, 356, 454, 323, 361, 288, 397, 365, 379, 378, 400]      # jaw contour
    + [46, 53, 52, 65, 55] + [285, 295, 282, 283, 276]                                        # eyebrows L / R
    + [6, 197, 195, 5] + [48, 115, 220, 45, 4]                                                # nose bridge / tip
    + [33, 160, 158, 133, 153, 144] + [362, 385, 387, 263, 373, 380]                          # eyes L / R
    + [61, 40, 37, 0, 267, 270, 291, 321, 314, 17, 84, 91]                                    # outer lip
    + [78, 82, 13, 312, 308, 317, 14, 87])                                                    # inner lip


class _SimpleFlameFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, shape, expr, rotation, jaw, translation):
        ins = [t.contiguous().float() for t in (shape, expr, rotation, jaw, translation)]
        for t in ins:
            if not t.is_cuda:
                raise L.OmfsError("SimpleFLAME runs in HIP kernels only: move the tensors to the GPU (no CPU path exists)")
        B = ins[0].shape[0]
        out = torch.empty(B, model.n_lmk, 3, device=ins[0].device)
        p = torch.empty_like(out)
        L.check(L.load().omfs_simpleflame_fwd(model._c, *[L.ptr(t) for t in ins], B, L.ptr(out), L.ptr(p), L.stream_ptr()),
                "omfs_simpleflame_fwd")
        ctx.model, ctx.B = model, B
        ctx.save_for_backward(ins[2], p)
        return out

    @staticmethod
    def backward(ctx, dout):
        rotation, p = ctx.saved_tensors
        m, B = ctx.model, ctx.B
        dev = dout.device
        dout = dout.contiguous().float()
        g = torch.empty(B, m.n_lmk, 3, device=dev)
        dshape, dexpr = torch.empty(B, m.n_shape, device=dev), torch.empty(B, m.n_expr, device=dev)
        drot, djaw, dtrans = (torch.empty(B, 3, device=dev) for _ in range(3))
        L.check(L.load().omfs_simpleflame_bwd(m._c, L.ptr(rotation), L.ptr(p), L.ptr(dout), B, L.ptr(g), L.ptr(dshape), L.ptr(dexpr),
                                              L.ptr(drot), L.ptr(djaw), L.ptr(dtrans), L.stream_ptr()), "omfs_simpleflame_bwd")
        return None, dshape, dexpr, drot, djaw, dtrans


class SimpleFLAME:
    """Minimal FLAME forward pass for landmark fitting (reference :69-197): linear shape and expression
    blendshapes, a jaw heuristic on the lower half of the face, one global rotation, translation,
    barycentric landmarks.  Callable like the reference's nn.Module."""

    def __init__(self, flame_model_path: str, n_shape: int = 100, n_expr: int = 50):
        self.n_shape, self.n_expr = n_shape, n_expr
        from .engine.flame_rig import load_flame_pickle
        model = load_flame_pickle(flame_model_path)          # latin1 pickle as in the reference; chumpy not required
        v_template = np.array(model["v_template"], dtype=np.float64)
        shapedirs = np.array(model["shapedirs"], dtype=np.float64)
        faces = np.array(model["f"], dtype=np.int64)
        lmk = np.load(str(FLAME_LMK_PATH), allow_pickle=True)[()]
        lmk_faces_idx = np.asarray(lmk["full_lmk_faces_idx"], np.int64).reshape(-1)
        bary = np.asarray(lmk["full_lmk_bary_coords"], np.float64).reshape(-1, 3)
        tri = faces[lmk_faces_idx]                                              # (L,3) vertex ids
        basis = np.concatenate([shapedirs[:, :, :n_shape], shapedirs[:, :, 300:300 + n_expr]], 2)   # (V,3,K)
        # the lower-face mask exactly as the reference forms it (fp32 tensor mean, :179): vertices that sit
        # on the mean height must fall on the same side
        vt32 = torch.tensor(np.array(model["v_template"], dtype=np.float32))
        lower = (vt32[:, 1] < vt32[:, 1].mean()).double().numpy()
        self.n_lmk = int(tri.shape[0])
        self.h_lmk_template = np.einsum("li,lic->lc", bary, v_template[tri]).astype(np.float32)
        self.h_lmk_basis = np.einsum("li,lick->lck", bary, basis[tri]).astype(np.float32)
        self.h_lmk_lower = np.einsum("li,li->l", bary, lower[tri]).astype(np.float32)
        self.lmk_faces_idx = torch.from_numpy(lmk_faces_idx)
        self.device = torch.device("cpu")
        self._c = None

    def to(self, device):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise L.OmfsError("SimpleFLAME runs in HIP kernels only: device must be a GPU (no CPU path exists)")
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.device)
        self._t, self._b, self._l = up(self.h_lmk_template), up(self.h_lmk_basis), up(self.h_lmk_lower)
        self._c = L.SimpleFlameC(self.n_lmk, self.n_shape, self.n_expr, L.ptr(self._t), L.ptr(self._b), L.ptr(self._l))
        return self

    def _axis_angle_to_matrix(self, axis_angle):
        from .engine.flame_rig import rodrigues
        return rodrigues(axis_angle)

    def forward(self, shape, expr, rotation, jaw, translation):
        """All batched (B, ...) -> landmarks (B, N_lmk, 3)."""
        if self._c is None:
            self.to(shape.device)
        return _SimpleFlameFn.apply(self, shape, expr, rotation, jaw, translation)

    __call__ = forward


def detect_landmarks_mediapipe(images_dir: str) -> list:
    """MediaPipe FaceMesh on every PNG -> list of (68,2) pixel arrays or None (reference :200-244).
    Third-party detector, not reimplemented: needs `mediapipe` and `cv2` at call time."""
    try:
        import cv2
        import mediapipe as mp
    except ImportError as e:
        raise ImportError("detect_landmarks_mediapipe needs the third-party packages mediapipe and opencv-python") from e
    mesh = mp.solutions.face_mesh.FaceMesh(static_image_mode=True, max_num_faces=1, refine_landmarks=True, min_detection_confidence=0.5)
    names = sorted(f for f in os.listdir(images_dir) if f.endswith(".png"))
    print(f"[flame_fitter] Detecting landmarks in {len(names)} frames …")
    found = []
    for name in names:
        img = cv2.imread(os.path.join(images_dir, name))
        res = mesh.process(cv2.cvtColor(img, cv2.COLOR_BGR2RGB)) if img is not None else None
        if res is None or not res.multi_face_landmarks:
            found.append(None)
            continue
        h, w = img.shape[:2]
        face = res.multi_face_landmarks[0].landmark
        found.append(np.array([[face[i].x * w, face[i].y * h] for i in MEDIAPIPE_TO_68], np.float32))
    mesh.close()
    print(f"[flame_fitter] Landmarks detected: {sum(l is not None for l in found)}/{len(names)}")
    return found


def estimate_head_pose_from_landmarks(landmarks_2d, image_size):
    """(pitch, yaw, roll) heuristics from 68 2-D landmarks in [-1,1] coordinates (reference :247-291):
    yaw = 1.5 (nose_x - jaw-centre_x), pitch = 0.5 (nose_y - eye_y), roll = 0.3 (eyeR_y - eyeL_y)."""
    if landmarks_2d is None:
        return 0.0, 0.0, 0.0
    W, H = image_size
    lmk = landmarks_2d.copy()
    lmk[:, 0] = lmk[:, 0] / W * 2 - 1
    lmk[:, 1] = lmk[:, 1] / H * 2 - 1
    n = len(lmk)
    nose = lmk[30] if n > 30 else lmk[n // 2]
    jaw_r = lmk[16] if n > 16 else lmk[-1]
    yaw = (nose[0] - (lmk[0][0] + jaw_r[0]) / 2) * 1.5
    pitch = roll = 0.0
    if n > 45:
        eye_l = lmk[36:42].mean(axis=0) if n > 42 else lmk[36]
        eye_r = lmk[42:48].mean(axis=0) if n > 48 else lmk[42]
        pitch = (nose[1] - (eye_l[1] + eye_r[1]) / 2) * 0.5
        roll = (eye_r[1] - eye_l[1]) * 0.3
    return float(pitch), float(yaw), float(roll)


def fit_flame_to_landmarks(landmarks_2d_list: list, image_size: tuple, flame_model_path: str, n_shape: int = 100,
                           n_expr: int = 50, lr: float = 0.01, n_iters: int = 200, device: str = "cuda") -> dict:
    """Adam fit of shape (shared), expr, rotation, jaw, translation to the 2-D landmarks of all frames at
    once (reference :294-444): pseudo-perspective x/(-z+1e-8), masked MSE, L2 regularisers 1e-3/1e-4/1e-3,
    temporal smoothness 1e-3, group learning rates lr*{0.1, 1, 0.3, 1, 0.5}; result padded to 300/100."""
    T = len(landmarks_2d_list)
    W, H = image_size
    valid = [i for i, l in enumerate(landmarks_2d_list) if l is not None]
    if not valid:
        raise ValueError("No faces detected in any frame.")
    flame = SimpleFLAME(flame_model_path, n_shape, n_expr).to(device)
    dev = flame.device

    print("[flame_fitter] Estimating initial head poses from landmarks...")
    init_rot = np.array([estimate_head_pose_from_landmarks(l, image_size) for l in landmarks_2d_list], np.float32).reshape(T, 3)
    print("[flame_fitter] Initial rotation range:")
    for name, col in (("Pitch", 0), ("Yaw", 1), ("Roll", 2)):
        print(f"  {name + ':':6s} {init_rot[:, col].min():.3f} to {init_rot[:, col].max():.3f}")

    n_pts = len(MEDIAPIPE_TO_68)
    target_h = np.zeros((T, n_pts, 2), np.float32)
    trans_h = np.zeros((T, 3), np.float32)
    trans_h[:, 2] = -5.0
    for i in valid:
        l = landmarks_2d_list[i]
        target_h[i, :, 0] = l[:, 0] / W * 2 - 1
        target_h[i, :, 1] = l[:, 1] / H * 2 - 1
        trans_h[i, 0] = float(l[:, 0].mean() / W * 2 - 1) * 2
        trans_h[i, 1] = float(l[:, 1].mean() / H * 2 - 1) * 2
    mask_h = np.zeros(T, bool)
    mask_h[valid] = True
    target, vmask = torch.from_numpy(target_h).to(dev), torch.from_numpy(mask_h).to(dev)

    # parameters, Adam moments and the whole iteration live on the device: one C-ABI call per iteration
    # (omfs_flame_fit_step: forward, loss, closed-form gradients incl. regularisers and smoothness, Adam)
    shape = torch.zeros(n_shape, device=dev)
    expr = torch.zeros(T, n_expr, device=dev)
    rotation = torch.from_numpy(init_rot).to(dev)
    jaw = torch.zeros(T, 3, device=dev)
    translation = torch.from_numpy(trans_h).to(dev)
    tensors = [shape, expr, rotation, jaw, translation]
    adam_m, adam_v = [torch.zeros_like(t) for t in tensors], [torch.zeros_like(t) for t in tensors]
    n_lmk = min(n_pts, flame.n_lmk)
    target_use = target[:, :n_lmk].contiguous()
    valid_f = vmask.float().contiguous()
    lib = L.load()
    scratch = torch.empty(int(lib.omfs_flame_fit_scratch_floats(flame._c, T)), device=dev)
    loss_dev = torch.zeros(1, device=dev)
    fit = L.FlameFitC()
    fit.n_frames, fit.n_use = T, n_lmk
    fit.target, fit.valid = L.ptr(target_use), L.ptr(valid_f)
    fit.inv_denom = 1.0 / max(int(mask_h.sum()) * n_lmk, 1)
    fit.lr[:] = [lr * 0.1, lr, lr * 0.3, lr, lr * 0.5]                # group learning rates (reference :356-362)
    fit.beta1, fit.beta2, fit.eps = 0.9, 0.999, 1e-8                   # torch.optim.Adam defaults
    fit.shape, fit.expr, fit.rotation, fit.jaw, fit.translation = (L.ptr(t) for t in tensors)
    for k in range(5):
        fit.m[k], fit.v[k] = L.ptr(adam_m[k]), L.ptr(adam_v[k])
    fit.scratch, fit.loss_out = L.ptr(scratch), L.ptr(loss_dev)

    print(f"[flame_fitter] Fitting FLAME to {len(valid)} frames …")
    stream = L.stream_ptr()
    for it in range(n_iters):
        fit.step = it + 1
        L.check(lib.omfs_flame_fit_step(flame._c, fit, stream), "omfs_flame_fit_step")
        if (it + 1) % 50 == 0:
            print(f"  iter {it + 1}/{n_iters} — loss: {loss_dev.item():.6f}")

    with torch.no_grad():
        final_rot = rotation.cpu().numpy()
        print("[flame_fitter] Final rotation range:")
        for name, col in (("Pitch", 0), ("Yaw", 1), ("Roll", 2)):
            print(f"  {name + ':':6s} {final_rot[:, col].min():.3f} to {final_rot[:, col].max():.3f}")
        shape_full = np.zeros(300, np.float32)
        shape_full[:n_shape] = shape.cpu().numpy()
        expr_full = np.zeros((T, 100), np.float32)
        expr_full[:, :n_expr] = expr.cpu().numpy()
        result = {"shape": shape_full, "expr": expr_full, "rotation": final_rot, "neck_pose": np.zeros((T, 3), np.float32),
                  "jaw_pose": jaw.cpu().numpy(), "eyes_pose": np.zeros((T, 6), np.float32), "translation": translation.cpu().numpy(),
                  "static_offset": np.zeros((1, 5143, 3), np.float32), "dynamic_offset": np.zeros((T, 5143, 3), np.float32)}
    print("[flame_fitter] Fitting complete.")
    return result


def fit_video(images_dir: str, output_path: str, device: str = "cuda", n_iters: int = 200):
    """detect landmarks -> fit -> np.savez (reference :447-479)."""
    if not FLAME_MODEL_PATH.exists():
        raise FileNotFoundError(f"FLAME model not found at: {FLAME_MODEL_PATH}\n"
                                "Copy flame2023.pkl to gaussian_avatars_repo/flame_model/assets/flame/")
    landmarks = detect_landmarks_mediapipe(images_dir)
    from .engine.io_formats import read_png
    first = sorted(f for f in os.listdir(images_dir) if f.endswith(".png"))[0]
    h, w = read_png(os.path.join(images_dir, first)).shape[:2]
    result = fit_flame_to_landmarks(landmarks, (w, h), str(FLAME_MODEL_PATH), n_iters=n_iters, device=device)
    np.savez(output_path, **result)
    print(f"[flame_fitter] Saved FLAME params: {output_path}")
    print(f"  Frames: {len(landmarks)}")
    print(f"  Shape params: {result['shape'].shape}")
    print(f"  Expr params:  {result['expr'].shape}")


def main():
    ap = argparse.ArgumentParser(description="Fit FLAME 2023 to video frames.")
    ap.add_argument("--images_dir", type=str, required=True)
    ap.add_argument("--output", type=str, required=True)
    ap.add_argument("--device", type=str, default="cuda")
    ap.add_argument("--n_iters", type=int, default=200)
    a = ap.parse_args()
    fit_video(a.images_dir, a.output, a.device, a.n_iters)


if __name__ == "__main__":
    main()

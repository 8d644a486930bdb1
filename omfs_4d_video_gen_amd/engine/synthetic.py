"""Seeded synthetic heads: FLAME-shaped rig, bound Gaussians, cameras, FLAME sequences.

The real FLAME weights are licensed and git-ignored by the reference
(`/root/reference/.gitignore:28-29`), so every test / bench input is generated here,
following SURVEY.md §8(d) "Synthetic inputs": an ellipsoid head with FLAME's 5023 vertices (top pole + 54 rings x 93
segments, open at the neck) to which the engine appends the 120 procedural teeth vertices of upstream's head
(`flame_rig.add_teeth`), V = 5143 as `static_offset (1,5143,3)` at `02_Visual_Engine/flame_fitter.py:439` says;
400 blendshape directions (300 shape + 100
expression, `flame_fitter.py:89-92`), 36 pose-corrective directions, and 5 joints
(root, neck, jaw, eye-L, eye-R; `flame_fitter.py:8-11`).

Everything is `numpy.random.default_rng(seed)`; nothing here touches the GPU.
"""
from __future__ import annotations

import math
import pickle
from dataclasses import dataclass

import numpy as np

V_BASE = 5023          # vertices of the FLAME pickle itself
N_TEETH = 120          # procedural teeth vertices appended by flame_rig.add_teeth (upstream's head has 5023 + 120)
V_FLAME = V_BASE + N_TEETH   # 5143: the vertex count of static_offset / dynamic_offset, flame_fitter.py:439-440
N_RINGS = 54
N_SEGS = 93
N_SHAPE = 300
N_EXPR = 100
N_POSEDIRS = 36
N_JOINTS = 5
PARENTS = np.array([-1, 0, 1, 1, 1], dtype=np.int64)
HEAD_RADII = (0.09, 0.12, 0.10)


def ellipsoid_mesh(radii=HEAD_RADII, rings=N_RINGS, segs=N_SEGS):
    """UV ellipsoid, open at the bottom like FLAME's neck: top pole + rings*segs vertices (1 + 54*93 = 5023),
    segs + 2*(rings-1)*segs faces (CCW seen from outside)."""
    rx, ry, rz = radii
    verts = [(0.0, ry, 0.0)]
    for r in range(rings):
        th = math.pi * (r + 1) / (rings + 1)
        for s in range(segs):
            ph = 2.0 * math.pi * s / segs
            verts.append((rx * math.sin(th) * math.cos(ph), ry * math.cos(th), rz * math.sin(th) * math.sin(ph)))
    verts = np.asarray(verts, dtype=np.float64)
    faces = []
    top = 0
    ring0 = lambda r: 1 + r * segs
    for s in range(segs):
        s1 = (s + 1) % segs
        faces.append((top, ring0(0) + s1, ring0(0) + s))
    for r in range(rings - 1):
        for s in range(segs):
            s1 = (s + 1) % segs
            a, b = ring0(r) + s, ring0(r) + s1
            c, d = ring0(r + 1) + s, ring0(r + 1) + s1
            faces.append((a, b, d))
            faces.append((a, d, c))
    return verts.astype(np.float32), np.asarray(faces, dtype=np.int32)


def lip_rings(rings=N_RINGS, segs=N_SEGS):
    """The two rows of 15 vertices the teeth are built from (upper lip, lower lip): adjacent rings a third of the way
    below the equator, 15 consecutive segments centred on the front (+z) meridian."""
    r_up = int(round(math.acos(-0.3) * (rings + 1) / math.pi)) - 1
    s0 = int(round(segs / 4.0)) - 7
    seg = np.arange(s0, s0 + 15)
    return (1 + r_up * segs + seg).astype(np.int64), (1 + (r_up + 1) * segs + seg).astype(np.int64)


@dataclass
class SyntheticRig:
    """Arrays in the FLAME pickle's own conventions (see `flame_rig.FlameRig.from_arrays`)."""
    v_template: np.ndarray      # (V,3) f32
    shapedirs: np.ndarray       # (V,3,400) f32   [:, :, :300] shape, [:, :, 300:] expression
    posedirs: np.ndarray        # (V,3,36) f32    (FLAME pickle layout)
    J_regressor: np.ndarray     # (5,V) f32, rows sum to 1
    weights: np.ndarray         # (V,5) f32, rows sum to 1
    kintree_table: np.ndarray   # (2,5) int64
    faces: np.ndarray           # (F,3) int32
    lmk_faces_idx: np.ndarray   # (68,) int64
    lmk_bary_coords: np.ndarray  # (68,3) f32
    n_base_verts: int = V_BASE  # the leading vertices / faces are what the FLAME pickle holds; the rest are the teeth
    n_base_faces: int = 0
    lip_upper: np.ndarray | None = None   # (15,) vertex ids the teeth were derived from
    lip_lower: np.ndarray | None = None


def _smooth_fields(rng, v: np.ndarray, k: int, std: float) -> np.ndarray:
    """(V,3,k) blendshape directions that vary smoothly over the surface (like FLAME's PCA bases;
    white noise per vertex would crumple the mesh): direction_k * cos(2 pi f_k . x / 0.1 + phi_k),
    per-vertex standard deviation `std` per unit coefficient."""
    f = rng.standard_normal((k, 3)) * 0.6
    phi = rng.uniform(0, 2 * math.pi, k)
    d = rng.standard_normal((k, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    phase = 2 * math.pi * (v.astype(np.float64) @ f.T) / 0.1 + phi[None, :]          # (V,k)
    field = np.cos(phase) * (std * math.sqrt(2.0) * math.sqrt(3.0))                     # unit-variance per axis on average
    return (field[:, None, :] * d.T[None, :, :]).astype(np.float32)


def make_rig(seed: int = 0, shape_std: float = 1e-3, pose_std: float = 1e-4) -> SyntheticRig:
    rng = np.random.default_rng(seed)
    v, f = ellipsoid_mesh()
    V = v.shape[0]
    assert V == V_BASE
    shapedirs = _smooth_fields(rng, v, N_SHAPE + N_EXPR, shape_std)
    posedirs = _smooth_fields(rng, v, N_POSEDIRS, pose_std)
    rx, ry, rz = HEAD_RADII
    joints = np.array([
        [0.0, -0.9 * ry, -0.2 * rz],     # root (base of neck)
        [0.0, -0.6 * ry, -0.1 * rz],     # neck
        [0.0, -0.2 * ry, 0.3 * rz],      # jaw hinge
        [0.35 * rx, 0.25 * ry, 0.8 * rz],   # eye L
        [-0.35 * rx, 0.25 * ry, 0.8 * rz],  # eye R
    ], dtype=np.float64)
    d2 = ((v[None, :, :].astype(np.float64) - joints[:, None, :]) ** 2).sum(-1)   # (5,V)
    # joint regressor: tight softmax around each joint location (rows sum to 1)
    jr = np.exp(-(d2 - d2.min(1, keepdims=True)) / (0.02 ** 2))
    jr /= jr.sum(1, keepdims=True)
    # skinning weights: soft assignment; eyes get tight support, jaw owns the lower front
    sig = np.array([0.08, 0.08, 0.05, 0.02, 0.02]) ** 2
    w = np.exp(-d2.T / sig[None, :])
    w[:, 0] += 1e-3
    w /= w.sum(1, keepdims=True)
    kintree = np.stack([np.array([2 ** 32 - 1, 0, 1, 1, 1], dtype=np.int64), np.arange(5, dtype=np.int64)])
    # 68 landmark triangles on the front (+z) half, seeded
    centers = v[f].mean(1)
    front = np.where(centers[:, 2] > 0.4 * rz)[0]
    lmk_f = np.sort(rng.choice(front, size=68, replace=False)).astype(np.int64)
    bary = rng.random((68, 3)) + 0.2
    bary = (bary / bary.sum(1, keepdims=True)).astype(np.float32)
    # the head the engine works on: the pickle's 5023 vertices + the 120 procedural teeth vertices (flame_rig.add_teeth)
    from .flame_rig import add_teeth
    up, low = lip_rings()
    t = add_teeth(v, shapedirs, posedirs, jr.astype(np.float32), w.astype(np.float32), f, up, low)
    return SyntheticRig(t["v_template"], t["shapedirs"], t["posedirs"], t["J_regressor"], t["weights"], kintree, t["faces"],
                        lmk_f, bary, V, f.shape[0], up, low)


def write_flame_pickle(rig: SyntheticRig, pkl_path: str, lmk_npy_path: str | None = None) -> None:
    """Write the rig with the key names `flame_fitter.SimpleFLAME.__init__` reads
    (`flame_fitter.py:80-120`): v_template, shapedirs, posedirs, J_regressor (object with
    `.todense()`), weights, kintree_table, f; landmark npy: dict with
    `full_lmk_faces_idx`, `full_lmk_bary_coords`."""
    import scipy.sparse as sp
    nv, nf = rig.n_base_verts, rig.n_base_faces     # the pickle holds FLAME itself: no teeth
    model = {
        "v_template": rig.v_template[:nv].astype(np.float64),
        "shapedirs": rig.shapedirs[:nv].astype(np.float64),
        "posedirs": rig.posedirs[:nv].astype(np.float64),
        "J_regressor": sp.csc_matrix(rig.J_regressor[:, :nv].astype(np.float64)),
        "weights": rig.weights[:nv].astype(np.float64),
        "kintree_table": rig.kintree_table,
        "f": rig.faces[:nf].astype(np.uint32),
    }
    with open(pkl_path, "wb") as fh:
        pickle.dump(model, fh, protocol=2)
    if lmk_npy_path:
        np.save(lmk_npy_path, {"full_lmk_faces_idx": rig.lmk_faces_idx,
                               "full_lmk_bary_coords": rig.lmk_bary_coords}, allow_pickle=True)


def make_gaussians(n: int, n_faces: int, seed: int = 0) -> dict:
    """Bound Gaussian cloud: binding i mod F; triangle-relative local frame."""
    rng = np.random.default_rng(seed + 1000)
    xyz = np.empty((n, 3), np.float32)
    # local frame axes are (edge, normal, in-plane) -- see DESIGN.md "face frame"
    xyz[:, 0] = rng.standard_normal(n) * 0.25
    xyz[:, 1] = rng.standard_normal(n) * 0.05
    xyz[:, 2] = rng.standard_normal(n) * 0.25
    log_scale = rng.uniform(math.log(0.2), math.log(1.0), (n, 3)).astype(np.float32)
    rot = rng.standard_normal((n, 4)).astype(np.float32)
    rot /= np.linalg.norm(rot, axis=1, keepdims=True)
    opacity = rng.uniform(-2.0, 4.0, n).astype(np.float32)
    sh = np.empty((n, 16, 3), np.float32)
    sh[:, 0, :] = rng.uniform(-1.0, 1.0, (n, 3))
    sh[:, 1:, :] = rng.standard_normal((n, 15, 3)) * 0.05
    binding = (np.arange(n, dtype=np.int64) % n_faces).astype(np.int32)
    return {"xyz": xyz, "log_scale": log_scale, "rot": rot, "opacity": opacity, "sh": sh, "binding": binding}


def make_camera(width: int, height: int, yaw: float = 0.0, fill: float = 0.6, distance: float = 1.0) -> dict:
    """Pinhole camera looking at the origin from +z rotated by `yaw` about y.

    Head (2*ry tall) fills `fill` of the image height. Convention: camera looks down its +z
    axis (3DGS / COLMAP), image x right, y down; `world_to_view` is 4x4 row-major.
    `camera_angle_x = 2 atan(w / (2 fl_x))` as in `preprocess_video.py:237`."""
    ry = HEAD_RADII[1]
    fl = fill * height * distance / (2.0 * ry)
    c, s = math.cos(yaw), math.sin(yaw)
    cam_pos = np.array([distance * s, 0.0, distance * c])
    zc = -cam_pos / np.linalg.norm(cam_pos)           # forward
    xc = np.cross(np.array([0.0, -1.0, 0.0]), zc)      # y-down camera => right = down x forward ... keep right-handed
    xc /= np.linalg.norm(xc)
    yc = np.cross(zc, xc)
    R = np.stack([xc, yc, zc])                         # rows: camera axes in world
    w2v = np.eye(4)
    w2v[:3, :3] = R
    w2v[:3, 3] = -R @ cam_pos
    return {
        "width": int(width), "height": int(height),
        "fl_x": float(fl), "fl_y": float(fl),
        "tanfovx": float(width / (2.0 * fl)), "tanfovy": float(height / (2.0 * fl)),
        "world_to_view": w2v.astype(np.float32),
        "cam_pos": cam_pos.astype(np.float32),
        "camera_angle_x": float(2.0 * math.atan(width / (2.0 * fl))),
    }


def make_camera_arc(width: int, height: int, n: int = 16, max_yaw_deg: float = 60.0, **kw) -> list:
    if n == 1:
        return [make_camera(width, height, 0.0, **kw)]
    yaws = np.linspace(-math.radians(max_yaw_deg), math.radians(max_yaw_deg), n)
    return [make_camera(width, height, float(y), **kw) for y in yaws]


def make_flame_sequence(T: int, seed: int = 0, identity: bool = False) -> dict:
    """FLAME parameter sequence in the dataset's npz schema (`flame_fitter.py:431-441`)."""
    rng = np.random.default_rng(seed + 2000)
    p = {
        "shape": np.zeros(N_SHAPE, np.float32),
        "expr": np.zeros((T, N_EXPR), np.float32),
        "rotation": np.zeros((T, 3), np.float32),
        "neck_pose": np.zeros((T, 3), np.float32),
        "jaw_pose": np.zeros((T, 3), np.float32),
        "eyes_pose": np.zeros((T, 6), np.float32),
        "translation": np.zeros((T, 3), np.float32),
        "static_offset": np.zeros((1, V_FLAME, 3), np.float32),
        "dynamic_offset": np.zeros((T, V_FLAME, 3), np.float32),
    }
    if identity:
        return p
    p["shape"] = (rng.standard_normal(N_SHAPE) * 0.5).astype(np.float32)
    e = rng.standard_normal((T + 16, N_EXPR)) * 0.5
    k = np.hanning(17)
    k /= k.sum()
    e = np.stack([np.convolve(e[:, j], k, mode="valid") for j in range(N_EXPR)], 1)[:T]
    p["expr"] = e.astype(np.float32)
    t = np.arange(T) / max(T - 1, 1)
    p["jaw_pose"][:, 0] = (0.15 * (1 - np.cos(2 * math.pi * 3 * t))).astype(np.float32)
    p["rotation"][:, 1] = (0.3 * np.sin(2 * math.pi * t)).astype(np.float32)
    p["neck_pose"][:, 0] = (0.05 * np.sin(2 * math.pi * 2 * t)).astype(np.float32)
    p["eyes_pose"][:, 1] = (0.1 * np.sin(2 * math.pi * 5 * t)).astype(np.float32)
    p["eyes_pose"][:, 4] = p["eyes_pose"][:, 1]
    p["translation"][:, 0] = (0.005 * np.sin(2 * math.pi * t)).astype(np.float32)
    p["static_offset"] = (rng.standard_normal((1, V_FLAME, 3)) * 2e-4).astype(np.float32)
    return p

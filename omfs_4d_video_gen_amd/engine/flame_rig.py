"""FLAME rig on the device: blendshapes + pose correctives + LBS -> vertices -> triangle frames.

Host side of `omfs_flame_joints / omfs_flame_lbs / omfs_face_frames` (include/omfs_splat.h).
The reference's own FLAME code (`02_Visual_Engine/flame_fitter.py:69-197`, SimpleFLAME) is
linear blendshapes + a jaw heuristic; the engine behind `train_ghost` / `render_surgery` needs
the full model, which is what this module drives.

`rodrigues` follows `SimpleFLAME._axis_angle_to_matrix` (`flame_fitter.py:122-152`) exactly and
is pinned by the golden vectors generated from the reference.
"""
from __future__ import annotations

import os
import pickle

import numpy as np
import torch

from .. import _lib as L

PARENTS = (-1, 0, 1, 1, 1)
N_SHAPE_MAX = 300
N_POSEDIRS = 36


def rodrigues(axis_angle: torch.Tensor) -> torch.Tensor:
    """(B,3) axis-angle -> (B,3,3); same formula and epsilon as flame_fitter.py:133-152."""
    B = axis_angle.shape[0]
    angle = torch.norm(axis_angle, dim=1, keepdim=True)
    axis = axis_angle / (angle + 1e-8)
    cos_a = torch.cos(angle).unsqueeze(-1)
    sin_a = torch.sin(angle).unsqueeze(-1)
    zeros = torch.zeros(B, dtype=axis_angle.dtype, device=axis_angle.device)
    K = torch.stack([
        torch.stack([zeros, -axis[:, 2], axis[:, 1]], dim=1),
        torch.stack([axis[:, 2], zeros, -axis[:, 0]], dim=1),
        torch.stack([-axis[:, 1], axis[:, 0], zeros], dim=1),
    ], dim=1)
    eye = torch.eye(3, dtype=axis_angle.dtype, device=axis_angle.device).unsqueeze(0).expand(B, -1, -1)
    return eye + sin_a * K + (1 - cos_a) * torch.bmm(K, K)


def pose_rotmats(flame_params: dict) -> np.ndarray:
    """Dataset FLAME params (`flame_fitter.py:431-441` schema) -> (T,5,3,3) float32
    [global rotation, neck, jaw, eye-L, eye-R]."""
    def as2d(a, w):
        a = np.asarray(a, np.float32)
        return a.reshape(-1, w)
    rot = as2d(flame_params["rotation"], 3)
    T = rot.shape[0]
    neck = as2d(flame_params.get("neck_pose", np.zeros((T, 3))), 3)
    jaw = as2d(flame_params.get("jaw_pose", np.zeros((T, 3))), 3)
    eyes = as2d(flame_params.get("eyes_pose", np.zeros((T, 6))), 6)
    aa = np.stack([rot, neck, jaw, eyes[:, :3], eyes[:, 3:]], 1).reshape(-1, 3)
    return rodrigues(torch.from_numpy(aa)).reshape(T, 5, 3, 3).numpy()


N_TEETH_COLS = 15      # vertices per teeth row (= vertices of a lip ring)
N_TEETH_VERTS = 8 * N_TEETH_COLS     # 120: FLAME's 5023 + 120 = the 5143 of static_offset (flame_fitter.py:439-440)
N_TEETH_FACES = 6 * 2 * (N_TEETH_COLS - 1)   # 168


def add_teeth(v_template, shapedirs, posedirs, J_regressor, weights, faces, lip_upper, lip_lower, inset: float = 0.004) -> dict:
    """Append upstream's procedural teeth to a FLAME rig: 120 vertices and 168 faces [NOT IN REFERENCE: the reference
    only fixes the vertex count, `static_offset (1,5143,3)` at flame_fitter.py:439; the construction follows the
    GaussianAvatars paper's description -- two arches derived from the lip rings, the upper one rigid with the head, the
    lower one rigid with the jaw].

    lip_upper / lip_lower: 15 vertex ids each, corresponding column by column.  With d = mean distance between the
    rings, the arches start from the rings' midline (height levelled, pushed `inset` metres into the mouth):
      rows 0..7 = upper root, lower root, upper edge, lower edge (front faces), then upper root, upper edge, lower root,
      lower edge again 0.5 d further back -- edges 0.1 d above / below the midline, roots 2 d beyond the edges, the
      lower arch another 0.1 d inside the upper one.
    Rig arrays of the new vertices: the SHAPE blendshapes (first 300 directions) of the lip vertex above / below, no
    expression and no pose-corrective displacement, no weight in the joint regressor, skinning weight 1 on the neck
    joint (index 1: the head) for the upper arch and on the jaw joint (index 2) for the lower arch.
    Faces: three quad strips per arch (front, biting edge, back), 14 quads each."""
    v = np.asarray(v_template, np.float32)
    V = v.shape[0]
    up, low = np.asarray(lip_upper, np.int64).reshape(-1), np.asarray(lip_lower, np.int64).reshape(-1)
    if up.shape != (N_TEETH_COLS,) or low.shape != (N_TEETH_COLS,):
        raise ValueError(f"teeth need two lip rings of {N_TEETH_COLS} vertices")
    vu, vl = v[up].astype(np.float64), v[low].astype(np.float64)
    d = float(np.linalg.norm(vu - vl, axis=1).mean())
    mid = 0.5 * (vu + vl)
    mid[:, 1] = mid[:, 1].mean()
    mid[:, 2] -= inset
    ey, ez = np.array([0.0, 1.0, 0.0]), np.array([0.0, 0.0, 1.0])
    u_edge = mid + 0.1 * d * ey
    u_root = u_edge + 2.0 * d * ey
    l_edge = mid - 0.1 * d * ey - 0.1 * d * ez
    l_root = l_edge - 2.0 * d * ey
    back = -0.5 * d * ez
    rows = [u_root, l_root, u_edge, l_edge, u_root + back, u_edge + back, l_root + back, l_edge + back]
    is_upper = [True, False, True, False, True, True, False, False]
    n_new = N_TEETH_VERTS
    sd, pd = np.asarray(shapedirs, np.float32), np.asarray(posedirs, np.float32)
    sd_new = np.zeros((n_new,) + sd.shape[1:], np.float32)
    w_new = np.zeros((n_new, 5), np.float32)
    for r, upper in enumerate(is_upper):
        sl = slice(r * N_TEETH_COLS, (r + 1) * N_TEETH_COLS)
        sd_new[sl, :, :N_SHAPE_MAX] = sd[up if upper else low][:, :, :N_SHAPE_MAX]
        w_new[sl, 1 if upper else 2] = 1.0
    row0 = lambda r: V + r * N_TEETH_COLS
    strips = [(0, 2), (2, 5), (5, 4),      # upper: front (root -> edge), biting edge (front -> back), back (edge -> root)
              (3, 1), (7, 3), (6, 7)]      # lower: front (edge -> root), biting edge (back -> front), back (root -> edge)
    f_new = []
    for ra, rb in strips:
        for c in range(N_TEETH_COLS - 1):
            a, b, cc, dd = row0(ra) + c, row0(ra) + c + 1, row0(rb) + c, row0(rb) + c + 1
            f_new += [(a, b, dd), (a, dd, cc)]
    jr = np.asarray(J_regressor, np.float32)
    return {
        "v_template": np.concatenate([v, np.concatenate(rows, 0).astype(np.float32)], 0),
        "shapedirs": np.concatenate([sd, sd_new], 0),
        "posedirs": np.concatenate([pd, np.zeros((n_new,) + pd.shape[1:], np.float32)], 0),
        "J_regressor": np.concatenate([jr, np.zeros((jr.shape[0], n_new), np.float32)], 1),
        "weights": np.concatenate([np.asarray(weights, np.float32), w_new], 0),
        "faces": np.concatenate([np.asarray(faces, np.int32), np.asarray(f_new, np.int32)], 0),
    }


def lip_rings_from_masks(masks_path: str, v_template: np.ndarray):
    """Lip rings for the released FLAME topology from the `lips` region of FLAME_masks.pkl (the file upstream ships next
    to the model): the region's vertices are split at the mouth's mean height, each half is ordered along x and cut into
    15 equally populated columns, and the front-most (largest z) vertex of each column is taken.  Returns (upper, lower)
    or None when a half holds fewer than 15 vertices.  [Could not be checked against the licensed asset in this build.]"""
    with open(masks_path, "rb") as f:
        masks = pickle.load(f, encoding="latin1")
    lips = np.asarray(masks["lips"], np.int64).reshape(-1)
    vv = np.asarray(v_template, np.float64)[lips]
    y_mid = vv[:, 1].mean()
    out = []
    for half in (vv[:, 1] >= y_mid, vv[:, 1] < y_mid):
        cand = np.nonzero(half)[0]
        if cand.size < N_TEETH_COLS:
            return None
        cand = cand[np.argsort(-vv[cand, 0], kind="stable")]          # +x first, like the synthetic rings
        out.append(np.asarray([int(lips[c[np.argmax(vv[c, 2])]]) for c in np.array_split(cand, N_TEETH_COLS)], np.int64))
    return out[0], out[1]


class FlameRig:
    """Static rig arrays (host, numpy) in FLAME-pickle conventions."""

    def __init__(self, v_template, shapedirs, posedirs, J_regressor, weights, faces, parents=PARENTS):
        self.v_template = np.asarray(v_template, np.float32)          # (V,3)
        self.shapedirs = np.asarray(shapedirs, np.float32)            # (V,3,>=300+n_expr)
        self.posedirs = np.asarray(posedirs, np.float32)              # (V,3,36)
        self.J_regressor = np.asarray(J_regressor, np.float32)        # (5,V)
        self.weights = np.asarray(weights, np.float32)                # (V,5)
        self.faces = np.asarray(faces, np.int32)                      # (F,3)
        if tuple(int(p) for p in parents) != PARENTS:
            raise ValueError(f"unsupported kinematic tree {tuple(parents)}; FLAME's is {PARENTS}")
        V = self.v_template.shape[0]
        if self.shapedirs.shape[:2] != (V, 3) or self.posedirs.shape != (V, 3, N_POSEDIRS):
            raise ValueError("rig array shapes do not match FLAME conventions")
        if self.J_regressor.shape != (5, V) or self.weights.shape != (V, 5):
            raise ValueError("FLAME has 5 joints: J_regressor (5,V), weights (V,5)")

    @property
    def n_verts(self):
        return self.v_template.shape[0]

    @property
    def n_faces(self):
        return self.faces.shape[0]

    @classmethod
    def from_synthetic(cls, rig):
        return cls(rig.v_template, rig.shapedirs, rig.posedirs, rig.J_regressor, rig.weights, rig.faces)

    @classmethod
    def from_pickle(cls, path: str, lip_rings=None, masks_path: str | None = None):
        """Same keys `SimpleFLAME.__init__` reads (`flame_fitter.py:80-120`) plus `posedirs`.
        The pickle holds FLAME's 5023 vertices; the 120 teeth vertices of upstream's head are appended (`add_teeth`)
        when the lip rings are known: given as `lip_rings` (upper ids, lower ids), or derived from `masks_path` /
        `FLAME_masks.pkl` next to the model.  Without them the rig keeps the pickle's vertex count (datasets carrying
        5143-row offsets then use the leading rows, see DeviceFlame)."""
        import os
        m = load_flame_pickle(path)
        jr = m["J_regressor"]
        jr = np.asarray(jr.todense() if hasattr(jr, "todense") else jr, np.float32)
        kt = np.asarray(m["kintree_table"], np.int64)
        parents = kt[0].copy()
        parents[0] = -1
        arrays = {"v_template": np.asarray(m["v_template"], np.float32), "shapedirs": np.asarray(m["shapedirs"], np.float32),
                  "posedirs": np.asarray(m["posedirs"], np.float32), "J_regressor": jr, "weights": np.asarray(m["weights"], np.float32),
                  "faces": np.asarray(m["f"], np.int64).astype(np.int32)}
        if lip_rings is None:
            cand = masks_path or os.path.join(os.path.dirname(os.path.abspath(path)), "FLAME_masks.pkl")
            if os.path.exists(cand):
                lip_rings = lip_rings_from_masks(cand, arrays["v_template"])
        if lip_rings is not None:
            arrays = add_teeth(**arrays, lip_upper=lip_rings[0], lip_lower=lip_rings[1])
        return cls(arrays["v_template"], arrays["shapedirs"], arrays["posedirs"], arrays["J_regressor"], arrays["weights"],
                   arrays["faces"], parents)


class _ChumpyArray:
    """Stands in for `chumpy.ch.Ch` (and relatives) when the FLAME pickle is read without chumpy installed: the released
    FLAME pickles store several arrays as chumpy objects whose state carries the ndarray under `x`."""

    def __setstate__(self, state):
        self.__dict__.update(state if isinstance(state, dict) else {"x": state})

    def __array__(self, dtype=None, copy=None):
        x = self.__dict__.get("x")
        if x is None:
            x = next((v for v in self.__dict__.values() if isinstance(v, np.ndarray)), None)
        if x is None:
            raise ValueError("chumpy object in the FLAME pickle carries no array")
        return np.asarray(x, dtype=dtype)


class _FlameUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module == "chumpy" or module.startswith("chumpy."):
            return _ChumpyArray
        return super().find_class(module, name)


def load_flame_pickle(path: str) -> dict:
    """`pickle.load(f, encoding="latin1")` as the reference does (`flame_fitter.py:80-82`), but independent of the
    chumpy package: with it installed the objects load as usual, without it they load as plain array holders."""
    with open(path, "rb") as f:
        try:
            return pickle.load(f, encoding="latin1")
        except ModuleNotFoundError as e:
            if "chumpy" not in str(e):
                raise
        f.seek(0)
        return _FlameUnpickler(f, encoding="latin1").load()


def tile_basis(basis_kcv: np.ndarray, k_pad: int, v_pad: int) -> np.ndarray:
    """[K][3][V] -> MFMA-A tiles [3][v_pad/16][k_pad/16][64][4]:
    lane l, element j holds row k = 16*kt + 4*j + (l>>4) of vertex strip*16 + (l&15)."""
    K, _, V = basis_kcv.shape
    pad = np.zeros((k_pad, 3, v_pad), np.float32)
    pad[:K, :, :V] = basis_kcv
    n_kt, n_strips = k_pad // 16, v_pad // 16
    t = pad.reshape(n_kt, 4, 4, 3, n_strips, 16)            # kt, j, grp, c, strip, vl
    return np.ascontiguousarray(t.transpose(3, 4, 0, 2, 5, 1)).reshape(3, n_strips, n_kt, 64, 4)


class DeviceFlame:
    """A rig specialised to one subject (shape + static_offset folded in) and one FLAME sequence,
    resident in HBM.  `face_frames(t)` runs the three FLAME kernels for frame(s) t."""

    def __init__(self, rig: FlameRig, flame_params: dict, device="cuda", n_expr: int | None = None):
        self.rig = rig
        self.device = torch.device(device)
        V, F = rig.n_verts, rig.n_faces
        expr = np.asarray(flame_params["expr"], np.float32)
        expr = expr.reshape(-1, expr.shape[-1])
        self.n_frames = expr.shape[0]
        self.n_expr = int(n_expr if n_expr is not None else expr.shape[1])
        if rig.shapedirs.shape[2] < N_SHAPE_MAX + self.n_expr:
            raise ValueError("rig has fewer expression directions than the sequence uses")
        self.v_pad = (V + 15) // 16 * 16
        K = self.n_expr + N_POSEDIRS
        self.k_pad = (K + 15) // 16 * 16
        shape = np.zeros(N_SHAPE_MAX, np.float64)
        s_in = np.asarray(flame_params["shape"], np.float64).reshape(-1)
        shape[:min(len(s_in), N_SHAPE_MAX)] = s_in[:N_SHAPE_MAX]
        v_static = rig.v_template.astype(np.float64) + rig.shapedirs[:, :, :N_SHAPE_MAX].astype(np.float64) @ shape
        so = flame_params.get("static_offset")
        if so is not None:
            so = np.asarray(so, np.float64).reshape(-1, 3)
            if so.shape[0] > V:
                # datasets written for upstream carry offsets for its procedural teeth (5023 FLAME + 120 appended
                # vertices, flame_fitter.py:439); a rig without them uses the leading V rows
                print(f"[engine] static_offset has {so.shape[0]} vertices, the rig {V}: using the first {V} (no teeth geometry)")
                so = so[:V]
            if so.shape[0] != V:
                raise ValueError(f"static_offset has {so.shape[0]} vertices, rig has {V}")
            v_static = v_static + so
        exprdirs = rig.shapedirs[:, :, N_SHAPE_MAX:N_SHAPE_MAX + self.n_expr].astype(np.float64)   # (V,3,E)
        basis = np.concatenate([exprdirs.transpose(2, 1, 0), rig.posedirs.astype(np.float64).transpose(2, 1, 0)], 0)
        JR = rig.J_regressor.astype(np.float64)
        # host copies (float32) of exactly what the kernels read -- the oracle gets the same arrays
        self.h_v_static = np.zeros((3, self.v_pad), np.float32)
        self.h_v_static[:, :V] = v_static.T.astype(np.float32)
        self.h_basis = basis.astype(np.float32)                                    # [K][3][V]
        self.h_j_static = (JR @ v_static).astype(np.float32)                       # (5,3)
        self.h_j_expr = np.einsum("jv,vck->jck", JR, exprdirs).reshape(15, self.n_expr).astype(np.float32)
        self.h_weights = np.zeros((self.v_pad, 8), np.float32)
        self.h_weights[:V, :5] = rig.weights
        self.h_rotmats = pose_rotmats(flame_params).reshape(self.n_frames, 45)
        self.h_expr = np.ascontiguousarray(expr[:, :self.n_expr])
        self.h_translation = np.asarray(flame_params["translation"], np.float32).reshape(-1, 3)
        dyn = flame_params.get("dynamic_offset")
        self.h_dynamic = None
        if dyn is not None and np.any(dyn):
            dyn = np.asarray(dyn, np.float32).reshape(self.n_frames, -1, 3)
            if dyn.shape[1] < V:
                raise ValueError(f"dynamic_offset has {dyn.shape[1]} vertices, rig has {V}")
            self.h_dynamic = np.ascontiguousarray(dyn[:, :V])
        dev = self.device
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        self.basis_tiled = up(tile_basis(self.h_basis, self.k_pad, self.v_pad))
        self.v_static = up(self.h_v_static)
        self.lbs_weights = up(self.h_weights)
        self.j_static = up(self.h_j_static)
        self.j_expr = up(self.h_j_expr)
        self.expr = up(self.h_expr)
        self.rotmats = up(self.h_rotmats)
        self.translation = up(self.h_translation)
        self.dynamic = up(self.h_dynamic) if self.h_dynamic is not None else None
        self.faces = up(rig.faces)
        self.c_rig = L.FlameRigC(V, self.v_pad, self.n_expr, self.k_pad, L.ptr(self.basis_tiled), L.ptr(self.v_static),
                                 L.ptr(self.lbs_weights), L.ptr(self.j_static), L.ptr(self.j_expr))
        self._scratch = {}
        self.keep_v_shaped = False     # FLAME fine-tuning: flame_lbs also stores the blend-shaped vertices
        self.pose = None               # FLAME fine-tuning: axis-angle poses [T][15]; rotmats are then refreshed by the joints launch
        self.slot = 0                  # output buffer set (the trainer poses the next step's frames on a side stream)

    def _buffers(self, nb: int):
        key = (nb, self.slot)
        if key not in self._scratch:
            b_pad = (nb + 15) // 16 * 16
            dev = self.device
            self._scratch[key] = (
                torch.empty(nb, 60, device=dev), torch.zeros(self.k_pad, b_pad, device=dev),
                torch.empty(nb, self.v_pad, 4, device=dev), torch.empty(nb, self.rig.n_faces, 16, device=dev),
                torch.empty(nb, self.v_pad, 4, device=dev) if self.keep_v_shaped else None)
        return self._scratch[key]

    def face_frames(self, t0: int, nb: int = 1, out=None):
        """Frames t0..t0+nb-1 -> (verts [nb][v_pad][4], face_xf [nb][F][16]) device tensors.
        The returned tensors are reused by the next call with the same nb."""
        if not (0 <= t0 and t0 + nb <= self.n_frames):
            raise IndexError(f"frames [{t0},{t0 + nb}) outside sequence of {self.n_frames}")
        return self._run(L.ptr(self.expr[t0]), L.ptr(self.rotmats[t0]), L.ptr(self.translation[t0]),
                         L.ptr(self.dynamic[t0]) if self.dynamic is not None else 0, nb, 0, out,
                         L.ptr(self.pose[t0]) if self.pose is not None else 0)

    def face_frames_indexed(self, frame_index: torch.Tensor):
        """Arbitrary timesteps in one batch: frame_index is a device int32 tensor [nb] of sequence rows.
        Returns (verts [nb][v_pad][4], face_xf [nb][F][16]); column b shows timestep frame_index[b]."""
        if frame_index.dtype != torch.int32 or not frame_index.is_cuda or frame_index.dim() != 1:
            raise ValueError("frame_index must be a 1-D int32 device tensor")
        return self._run(L.ptr(self.expr), L.ptr(self.rotmats), L.ptr(self.translation),
                         L.ptr(self.dynamic) if self.dynamic is not None else 0, int(frame_index.shape[0]), L.ptr(frame_index), None,
                         L.ptr(self.pose))

    def _run(self, expr_p, rot_p, trans_p, dyn_p, nb, index_p, out, pose_p=0):
        lib = L.load()
        s = L.stream_ptr()
        joint_xf, coef, verts, face_xf, v_shaped = self._buffers(nb)
        if out is not None:
            face_xf = out
        if nb == 1 and os.environ.get("OMFS_FLAME_SPLIT", "0") != "1":
            # one frame (the training step): joints and skinning in ONE launch, every wave evaluating the joints itself
            L.check(lib.omfs_flame_pose_lbs(self.c_rig, expr_p, rot_p, pose_p, trans_p, dyn_p, L.ptr(joint_xf), L.ptr(coef), L.ptr(verts),
                                            L.ptr(v_shaped), index_p, s), "omfs_flame_pose_lbs")
            L.check(lib.omfs_face_frames(L.ptr(verts), self.v_pad, L.ptr(self.faces), self.rig.n_faces, nb, L.ptr(face_xf), s),
                    "omfs_face_frames")
            return verts, face_xf
        if pose_p:     # rotation matrices from the current axis-angle poses, in the same launch
            L.check(lib.omfs_flame_joints_pose(self.c_rig, expr_p, pose_p, rot_p, nb, L.ptr(joint_xf), L.ptr(coef), index_p, s),
                    "omfs_flame_joints_pose")
        else:
            L.check(lib.omfs_flame_joints(self.c_rig, expr_p, rot_p, nb, L.ptr(joint_xf), L.ptr(coef), index_p, s), "omfs_flame_joints")
        L.check(lib.omfs_flame_lbs(self.c_rig, L.ptr(coef), L.ptr(joint_xf), trans_p, dyn_p, nb, L.ptr(verts), L.ptr(v_shaped),
                                   index_p, s), "omfs_flame_lbs")
        L.check(lib.omfs_face_frames(L.ptr(verts), self.v_pad, L.ptr(self.faces), self.rig.n_faces, nb, L.ptr(face_xf), s),
                "omfs_face_frames")
        return verts, face_xf

"""ORACLE (test infrastructure): ctypes wrapper of oracle/splat_oracle.c (bit-level forward spec).
Build with `make -C oracle`.  See the C file's header for scope and the "parity unpinned" note."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libsplat_oracle.so")
_lib = None


class OrcCamera(C.Structure):
    _fields_ = [("view", C.c_float * 12), ("cam_pos", C.c_float * 3), ("fx", C.c_float), ("fy", C.c_float),
                ("cx", C.c_float), ("cy", C.c_float), ("limx", C.c_float), ("limy", C.c_float),
                ("width", C.c_int), ("height", C.c_int), ("sh_degree", C.c_int), ("bg", C.c_float * 3)]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.orc_exp.restype = C.c_float
        _lib.orc_exp.argtypes = [C.c_float]
        _lib.orc_bin_sort.restype = C.c_int64
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def f32(a):
    return np.ascontiguousarray(a, np.float32)


def camera(c) -> OrcCamera:
    """from the engine's CameraC (field-compatible)"""
    o = OrcCamera()
    for name, _ in OrcCamera._fields_:
        v = getattr(c, name)
        if hasattr(v, "__len__"):
            getattr(o, name)[:] = list(v)
        else:
            setattr(o, name, v)
    return o


def exp_exact(x: np.ndarray) -> np.ndarray:
    l = lib()
    return np.array([l.orc_exp(float(v)) for v in np.asarray(x, np.float32).reshape(-1)], np.float32).reshape(np.shape(x))


def flame_frame(dflame, t: int):
    """dflame: engine DeviceFlame (only its host-side h_* arrays are read). -> verts (V,3), joint_xf (5,12)"""
    V = dflame.rig.n_verts
    verts = np.zeros((V, 3), np.float32)
    jx = np.zeros((5, 12), np.float32)
    dyn = f32(dflame.h_dynamic[t]) if dflame.h_dynamic is not None else None
    lib().orc_flame_frame(C.c_int(V), C.c_int(dflame.v_pad), C.c_int(dflame.n_expr), _p(f32(dflame.h_v_static)),
                          _p(f32(dflame.h_basis)), _p(f32(dflame.h_weights)), _p(f32(dflame.h_j_static)),
                          _p(f32(dflame.h_j_expr)), _p(f32(dflame.h_expr[t])), _p(f32(dflame.h_rotmats[t])),
                          _p(f32(dflame.h_translation[t])), _p(dyn), _p(verts), _p(jx))
    return verts, jx


def face_frames(verts: np.ndarray, faces: np.ndarray) -> np.ndarray:
    F = faces.shape[0]
    out = np.zeros((F, 16), np.float32)
    lib().orc_face_frames(C.c_int(F), _p(f32(verts)), _p(np.ascontiguousarray(faces, np.int32)), _p(out))
    return out


def project(params: np.ndarray, binding: np.ndarray, face_xf: np.ndarray, cam: OrcCamera, n: int):
    n_pad = params.shape[1]
    o = {"mean2d": np.zeros((n, 2), np.float32), "conic": np.zeros((n, 3), np.float32), "opac": np.zeros(n, np.float32),
         "rgb": np.zeros((n, 3), np.float32), "depth": np.zeros(n, np.float32), "radius": np.zeros(n, np.int32),
         "rect": np.zeros((n, 4), np.int32), "clamp": np.zeros(n, np.int32)}
    lib().orc_project(C.c_int(n), C.c_int(n_pad), _p(f32(params)), _p(np.ascontiguousarray(binding, np.int32)),
                      _p(f32(face_xf)), C.byref(cam), _p(o["mean2d"]), _p(o["conic"]), _p(o["opac"]), _p(o["rgb"]),
                      _p(o["depth"]), _p(o["radius"]), _p(o["rect"]), _p(o["clamp"]))
    return o


def bin_sort(proj: dict, width: int, height: int, cull: bool = True):
    """cull=True: the engine's rule (frozen tile test); cull=False: every tile of the 3-sigma rectangle."""
    n = proj["depth"].shape[0]
    nt = ((width + 15) // 16) * ((height + 15) // 16)
    tile_start = np.zeros(nt + 1, np.uint32)
    args = (C.c_int(n), C.c_int(width), C.c_int(height), _p(proj["depth"]), _p(proj["radius"]), _p(proj["rect"]),
            _p(proj["mean2d"]), _p(proj["conic"]), _p(proj["opac"]), C.c_int(1 if cull else 0), _p(tile_start))
    D = lib().orc_bin_sort(*args, None)
    ids = np.zeros(max(int(D), 1), np.uint32)
    lib().orc_bin_sort(*args, _p(ids))
    return tile_start, ids[:int(D)]


NEAR_TOL_ALPHA = 2e-5   # |255 alpha - 1| below which a (Gaussian, pixel) pair counts as "on the 1/255 threshold" (= torch_splat.NEAR_TOL)
NEAR_TOL_T = 2e-5       # relative distance of T(1 - alpha) from the 1e-4 stop threshold below which a pixel's stop is "on the threshold"


def composite(proj: dict, tile_start, ids, width: int, height: int, bg, near: bool = False):
    """near=True: also returns the [H][W] uint8 map of pixels that own a decision on a threshold (bit 0: alpha vs 1/255 within
    NEAR_TOL_ALPHA, bit 1: T vs 1e-4 within NEAR_TOL_T) -- see orc_composite_diag."""
    img = np.zeros((3, height, width), np.float32)
    fT = np.zeros((height, width), np.float32)
    nc = np.zeros((height, width), np.uint32)
    ids_ = np.ascontiguousarray(ids if len(ids) else np.zeros(1, np.uint32), np.uint32)
    args = (C.c_int(width), C.c_int(height), _p(f32(bg)), _p(tile_start), _p(ids_), _p(proj["mean2d"]),
            _p(proj["conic"]), _p(proj["opac"]), _p(proj["rgb"]), _p(img), _p(fT), _p(nc))
    if not near:
        lib().orc_composite(*args)
        return img, fT, nc
    flags = np.zeros((height, width), np.uint8)
    lib().orc_composite_diag(*args, _p(flags), C.c_float(NEAR_TOL_ALPHA), C.c_float(NEAR_TOL_T))
    return img, fT, nc, flags


def render(dflame, t, params, binding, n, cam: OrcCamera, cull: bool = True):
    """Whole forward for one frame: returns dict(verts, face_xf, proj, tile_start, ids, image, final_T, n_contrib)."""
    verts, jx = flame_frame(dflame, t)
    fxf = face_frames(verts, dflame.rig.faces)
    proj = project(params, binding, fxf, cam, n)
    ts, ids = bin_sort(proj, cam.width, cam.height, cull)
    img, fT, nc, near = composite(proj, ts, ids, cam.width, cam.height, list(cam.bg), near=True)
    return {"verts": verts, "joint_xf": jx, "face_xf": fxf, "proj": proj, "tile_start": ts, "ids": ids,
            "image": img, "final_T": fT, "n_contrib": nc, "near": near}

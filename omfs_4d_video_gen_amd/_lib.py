"""ctypes binding of libomfs_splat.so (C ABI declared in include/omfs_splat.h).

There is deliberately no CPU fallback: if the HIP library is missing or a call fails the
engine raises.  `torch` is imported first so that the library binds to the HIP runtime that
PyTorch-ROCm has already loaded (same SONAME, libamdhip64.so.7) and device pointers of torch
tensors are valid in our kernels.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must be loaded before the HIP library, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
# OMFS_LIB_PATH: another build of the SAME ABI (A/B variants under tools/_ab/so/): the tools point the binding at a candidate
# instead of overwriting the in-tree library, so that nothing run afterwards silently describes a stale variant.
LIB_PATH = os.environ.get("OMFS_LIB_PATH") or os.path.join(_HERE, "libomfs_splat.so")
ABI_VERSION = 8
LOSS_TAIL = 16384              # OMFS_LOSS_TAIL: loss partials behind the three maps of omfs_loss_l1_ssim's scratch
RB_FORWARD_ONLY = 1
NPLANES = 59
TILE = 16
SEG = 128

c_float_p = C.POINTER(C.c_float)
c_u32_p = C.POINTER(C.c_uint32)
c_i32_p = C.POINTER(C.c_int32)
c_void_p = C.c_void_p


class OmfsError(RuntimeError):
    pass


class FlameRigC(C.Structure):
    _fields_ = [("n_verts", C.c_int), ("v_pad", C.c_int), ("n_expr", C.c_int), ("k_pad", C.c_int),
                ("basis_tiled", c_void_p), ("v_static", c_void_p), ("lbs_weights", c_void_p),
                ("j_static", c_void_p), ("j_expr", c_void_p)]


class SimpleFlameC(C.Structure):
    _fields_ = [("n_landmarks", C.c_int), ("n_shape", C.c_int), ("n_expr", C.c_int),
                ("lmk_template", c_void_p), ("lmk_basis", c_void_p), ("lmk_lower", c_void_p)]


class CameraC(C.Structure):
    _fields_ = [("view", C.c_float * 12), ("cam_pos", C.c_float * 3), ("fx", C.c_float), ("fy", C.c_float),
                ("cx", C.c_float), ("cy", C.c_float), ("limx", C.c_float), ("limy", C.c_float),
                ("width", C.c_int), ("height", C.c_int), ("sh_degree", C.c_int), ("bg", C.c_float * 3)]


class GaussiansC(C.Structure):
    _fields_ = [("n", C.c_int), ("n_pad", C.c_int), ("params", c_void_p), ("binding", c_void_p)]


class RasterBuffersC(C.Structure):
    _fields_ = [("g0", c_void_p), ("g1", c_void_p), ("g2", c_void_p),
                ("tile_count", c_void_p), ("tile_start", c_void_p), ("tile_cursor", c_void_p),
                ("tile_order", c_void_p), ("keys", c_void_p), ("keys_tmp", c_void_p), ("sorted_ids", c_void_p),
                ("dup_capacity", C.c_uint32), ("sort_lds_pairs", C.c_uint32), ("status", c_void_p),
                ("seg_ckpt", c_void_p), ("order_seg0", c_void_p), ("seg_capacity", C.c_uint32),
                ("image", c_void_p), ("final_T", c_void_p), ("n_contrib", c_void_p), ("flags", C.c_uint32),
                ("n_visible", c_void_p), ("quad_depth", c_void_p)]


class GradBuffersC(C.Structure):
    _fields_ = [("dsplat", c_void_p), ("grads", c_void_p), ("dimage", c_void_p), ("densify_stats", c_void_p),
                ("dface", c_void_p), ("drgb_out", c_void_p), ("dir_out", c_void_p), ("dsplat_fx", c_void_p), ("n_records", C.c_uint32)]


class ViewSetC(C.Structure):
    _fields_ = [("n_views", C.c_int), ("view", C.c_int * 16)]


class RegParamsC(C.Structure):
    _fields_ = [("lambda_xyz", C.c_float), ("thr_xyz", C.c_float), ("lambda_scale", C.c_float),
                ("thr_scale", C.c_float), ("n_visible", c_void_p)]


class AdamParamsC(C.Structure):
    _fields_ = [("lr", C.c_float * NPLANES), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("step", C.c_int), ("grad_scale", C.c_float)]


class ViewStepC(C.Structure):
    _fields_ = [("g", C.POINTER(GaussiansC)), ("face_xf", c_void_p), ("cam", C.POINTER(CameraC)), ("rb", C.POINTER(RasterBuffersC)),
                ("gb", C.POINTER(GradBuffersC)), ("reg", C.POINTER(RegParamsC)), ("target", c_void_p), ("target_rgb8", c_void_p),
                ("target_scratch", c_void_p), ("lambda_dssim", C.c_float), ("loss_out", c_void_p), ("loss_scratch", c_void_p)]


class LrScheduleC(C.Structure):
    _fields_ = [("lr_init", C.c_float), ("lr_final", C.c_float), ("max_steps", C.c_int), ("beta1", C.c_float), ("beta2", C.c_float)]


class StepStateC(C.Structure):
    """omfs_step_state lives in DEVICE memory (the trainer keeps it as an int32 tensor); this mirror states its layout."""
    _fields_ = [("step", C.c_int32), ("flame_step", C.c_int32), ("lr_xyz", C.c_float), ("inv_bc1", C.c_float),
                ("inv_sqrt_bc2", C.c_float), ("flame_inv_bc1", C.c_float), ("flame_inv_sqrt_bc2", C.c_float),
                ("table_base", C.c_int32), ("reserved", C.c_float * 8)]


STEP_STATE_WORDS = C.sizeof(StepStateC) // 4
STEP_STATE_TABLE_BASE = StepStateC.table_base.offset // 4


class FlameFitC(C.Structure):
    _fields_ = [("n_frames", C.c_int), ("n_use", C.c_int), ("target", c_void_p), ("valid", c_void_p), ("inv_denom", C.c_float),
                ("lr", C.c_float * 5), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float), ("step", C.c_int),
                ("shape", c_void_p), ("expr", c_void_p), ("rotation", c_void_p), ("jaw", c_void_p), ("translation", c_void_p),
                ("m", c_void_p * 5), ("v", c_void_p * 5), ("scratch", c_void_p), ("loss_out", c_void_p)]


class DensifyParamsC(C.Structure):
    _fields_ = [("grad_threshold", C.c_float), ("size_threshold", C.c_float), ("min_opacity", C.c_float),
                ("prune_size", C.c_float), ("seed_lo", C.c_uint32), ("seed_hi", C.c_uint32)]


# name -> (restype, argtypes); this table is also what tests/test_abi.py checks against the header
SIGNATURES = {
    "omfs_abi_version": (C.c_int, []),
    "omfs_last_error": (C.c_char_p, []),
    "omfs_flame_joints": (C.c_int, [C.POINTER(FlameRigC), c_void_p, c_void_p, C.c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "omfs_flame_joints_pose": (C.c_int, [C.POINTER(FlameRigC), c_void_p, c_void_p, c_void_p, C.c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "omfs_adam_flat_multi": (C.c_int, [C.c_int, C.POINTER(c_void_p), C.POINTER(c_void_p), C.POINTER(c_void_p), C.POINTER(c_void_p),
                                       C.POINTER(C.c_int), C.POINTER(C.c_float), C.c_float, C.c_float, C.c_float, C.c_int, C.c_float,
                                       c_void_p, c_void_p]),
    "omfs_adam_step_range": (C.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, C.c_int, C.c_longlong, C.c_longlong,
                                       C.POINTER(AdamParamsC), c_void_p]),
    "omfs_view_forward_backward": (C.c_int, [C.POINTER(ViewStepC), c_void_p]),
    "omfs_view_forward_composite_bwd": (C.c_int, [C.POINTER(ViewStepC), c_void_p, c_void_p]),
    "omfs_view_project_bwd": (C.c_int, [C.POINTER(ViewStepC), c_void_p]),
    "omfs_step_advance": (C.c_int, [c_void_p, C.POINTER(LrScheduleC), c_void_p, C.c_int, c_void_p, c_void_p]),
    "omfs_adam_step_dev": (C.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, C.c_int, C.c_int, C.POINTER(AdamParamsC), c_void_p,
                                     C.c_int, C.c_int, c_void_p]),
    "omfs_flame_lbs": (C.c_int, [C.POINTER(FlameRigC), c_void_p, c_void_p, c_void_p, c_void_p, C.c_int, c_void_p, c_void_p,
                                 c_void_p, c_void_p]),
    "omfs_extract_drgb": (C.c_int, [C.POINTER(RasterBuffersC), c_void_p, C.c_int, C.c_int, c_void_p, c_void_p]),
    "omfs_comm_unique_id": (C.c_int, [c_void_p]),
    "omfs_comm_create": (C.c_int, [c_void_p, C.c_int, C.c_int, C.POINTER(c_void_p)]),
    "omfs_comm_destroy": (C.c_int, [c_void_p]),
    "omfs_rccl_allreduce_grads": (C.c_int, [c_void_p, c_void_p, C.c_size_t, c_void_p]),
    "omfs_rccl_allgather": (C.c_int, [c_void_p, c_void_p, c_void_p, C.c_size_t, c_void_p]),
    "omfs_rccl_reduce_scatter": (C.c_int, [c_void_p, c_void_p, c_void_p, C.c_size_t, c_void_p]),
    "omfs_sh_rest_grads": (C.c_int, [C.POINTER(GaussiansC), c_void_p, C.c_int, c_void_p, C.POINTER(ViewSetC), c_void_p, C.c_int,
                                     c_void_p, c_void_p]),
    "omfs_face_frames": (C.c_int, [c_void_p, C.c_int, c_void_p, C.c_int, C.c_int, c_void_p, c_void_p]),
    "omfs_face_frames_bwd": (C.c_int, [c_void_p, C.c_int, c_void_p, C.c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "omfs_face_frames_bwd_fx": (C.c_int, [c_void_p, C.c_int, c_void_p, C.c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "omfs_flame_skin_bwd": (C.c_int, [C.POINTER(FlameRigC), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "omfs_flame_rodrigues": (C.c_int, [c_void_p, C.c_int, c_void_p, c_void_p]),
    "omfs_flame_param_bwd": (C.c_int, [C.POINTER(FlameRigC), c_void_p, C.c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_void_p, c_void_p]),
    "omfs_flame_skin_rows": (C.c_int, [C.POINTER(FlameRigC)]),
    "omfs_flame_skin_param_scratch_floats": (C.c_int, [C.c_int]),
    "omfs_flame_skin_param_bwd": (C.c_int, [C.POINTER(FlameRigC), c_void_p, C.c_int] + [c_void_p] * 11),
    "omfs_flame_pose_lbs": (C.c_int, [C.POINTER(FlameRigC)] + [c_void_p] * 11),
    "omfs_adam_flat": (C.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float,
                                 C.c_int, C.c_float, c_void_p]),
    "omfs_simpleflame_fwd": (C.c_int, [C.POINTER(SimpleFlameC)] + [c_void_p] * 5 + [C.c_int, c_void_p, c_void_p, c_void_p]),
    "omfs_simpleflame_bwd": (C.c_int, [C.POINTER(SimpleFlameC)] + [c_void_p] * 3 + [C.c_int] + [c_void_p] * 7),
    "omfs_record_stride": (C.c_int, []),
    "omfs_project_fwd": (C.c_int, [C.POINTER(GaussiansC), c_void_p, C.POINTER(CameraC), C.POINTER(RasterBuffersC), c_void_p]),
    "omfs_bin_count": (C.c_int, [C.POINTER(GaussiansC), C.POINTER(CameraC), C.POINTER(RasterBuffersC), c_void_p]),
    "omfs_bin_scan": (C.c_int, [C.POINTER(CameraC), C.POINTER(RasterBuffersC), c_void_p]),
    "omfs_bin_scatter": (C.c_int, [C.POINTER(GaussiansC), C.POINTER(CameraC), C.POINTER(RasterBuffersC), c_void_p]),
    "omfs_tile_sort": (C.c_int, [C.POINTER(CameraC), C.POINTER(RasterBuffersC), c_void_p]),
    "omfs_bin_sort": (C.c_int, [C.POINTER(GaussiansC), C.POINTER(CameraC), C.POINTER(RasterBuffersC), c_void_p]),
    "omfs_composite_fwd": (C.c_int, [C.POINTER(CameraC), C.POINTER(RasterBuffersC), c_void_p]),
    "omfs_image_to_rgb8": (C.c_int, [c_void_p, C.c_int, C.c_int, c_void_p, c_void_p]),
    "omfs_image_to_png_rows": (C.c_int, [c_void_p, C.c_int, C.c_int, c_void_p, c_void_p]),
    "omfs_rgb8_to_image": (C.c_int, [c_void_p, C.c_int, C.c_int, c_void_p, c_void_p]),
    "omfs_png_slot_stride": (C.c_int, [C.c_int]),
    "omfs_png_deflate": (C.c_int, [c_void_p, C.c_int, C.c_int, c_void_p, c_void_p, c_void_p, c_void_p, C.c_uint32, c_void_p, c_void_p]),
    "omfs_png_fetch": (C.c_longlong, [c_void_p, c_void_p, C.c_size_t, C.c_size_t, c_void_p, c_void_p]),
    "omfs_prepare_target": (C.c_int, [c_void_p, C.c_int, C.c_int, C.c_int, c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float), c_void_p,
                                      c_void_p, c_void_p]),
    "omfs_composite_bwd": (C.c_int, [C.POINTER(CameraC), C.POINTER(RasterBuffersC), C.POINTER(GradBuffersC), c_void_p]),
    "omfs_project_bwd": (C.c_int, [C.POINTER(GaussiansC), c_void_p, C.POINTER(CameraC), C.POINTER(RasterBuffersC),
                                   C.POINTER(GradBuffersC), C.POINTER(RegParamsC), c_void_p]),
    "omfs_loss_l1_ssim": (C.c_int, [c_void_p, c_void_p, C.c_int, C.c_int, C.c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
    "omfs_adam_step": (C.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, C.c_int, C.c_int, C.POINTER(AdamParamsC), c_void_p]),
    "omfs_adam_step_planes": (C.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, C.c_int, C.c_int, C.POINTER(AdamParamsC), C.c_int, C.c_int,
                                        c_void_p]),
    "omfs_adam_step_sh_rest": (C.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, C.c_int, C.c_int, C.POINTER(AdamParamsC),
                                         C.c_int, c_void_p]),
    "omfs_count_visible": (C.c_int, [C.POINTER(RasterBuffersC), C.c_int, c_void_p, c_void_p]),
    "omfs_flame_fit_scratch_floats": (C.c_size_t, [C.POINTER(SimpleFlameC), C.c_int]),
    "omfs_flame_fit_step": (C.c_int, [C.POINTER(SimpleFlameC), C.POINTER(FlameFitC), c_void_p]),
    "omfs_densify_classify": (C.c_int, [C.POINTER(GaussiansC), c_void_p, c_void_p, C.POINTER(DensifyParamsC), c_void_p, c_void_p,
                                        c_void_p, c_void_p]),
    "omfs_densify_scan": (C.c_int, [c_void_p, C.c_int, c_void_p, c_void_p]),
    "omfs_densify_compact": (C.c_int, [C.POINTER(GaussiansC), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       C.POINTER(DensifyParamsC), C.c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
}

_lib = None


def load() -> C.CDLL:
    """Load the library (once).  Raises OmfsError if it is not built: no fallback exists."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OmfsError(f"{LIB_PATH} not found: build it with omfs_4d_video_gen_amd/csrc/build.sh "
                        "(or __graft_entry__.build()); the engine has no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    v = lib.omfs_abi_version()
    if v != ABI_VERSION:
        raise OmfsError(f"ABI version mismatch: library {v}, binding {ABI_VERSION}")
    _lib = lib
    return lib


EXPERIMENTS_PATH = os.path.join(_HERE, "libomfs_experiments.so")
BWD_IMPLS = ("dpp", "mfma", "entries")     # "dpp" is the product's omfs_composite_bwd; the others live in libomfs_experiments.so
_exp = None


def load_experiments() -> C.CDLL:
    """Second implementations of the composite backward pass (csrc/composite_experiments.hip): tests and tools only --
    nothing under engine/ calls this."""
    global _exp
    if _exp is None:
        if not os.path.exists(EXPERIMENTS_PATH):
            raise OmfsError(f"{EXPERIMENTS_PATH} not found: build it with omfs_4d_video_gen_amd/csrc/build.sh")
        load()
        lib = C.CDLL(EXPERIMENTS_PATH)
        lib.omfs_experiment_composite_bwd.restype = C.c_int
        lib.omfs_experiment_composite_bwd.argtypes = [C.c_char_p, C.POINTER(CameraC), C.POINTER(RasterBuffersC), C.POINTER(GradBuffersC), c_void_p]
        lib.omfs_experiment_last_error.restype = C.c_char_p
        _exp = lib
    return _exp


def composite_bwd(impl: str, cam, rb, gb, stream) -> None:
    """omfs_composite_bwd ("dpp") or one of the experimental implementations of the same contract, by name."""
    if impl == "dpp":
        check(load().omfs_composite_bwd(cam, rb, gb, stream), "omfs_composite_bwd")
        return
    lib = load_experiments()
    rc = lib.omfs_experiment_composite_bwd(impl.encode(), cam, rb, gb, stream)
    if rc != 0:
        raise OmfsError(f"omfs_experiment_composite_bwd({impl}) failed ({rc}): {lib.omfs_experiment_last_error().decode()}")


def library_identity() -> dict:
    """Which library this process runs: path, ABI version, sha256 prefix of the file (bench.py prints it into its line)."""
    import hashlib
    load()
    with open(LIB_PATH, "rb") as f:
        h = hashlib.sha256(f.read()).hexdigest()[:16]
    return {"path": os.path.relpath(LIB_PATH, os.path.dirname(_HERE)), "abi_version": ABI_VERSION, "sha256_16": h}


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise OmfsError(f"{what} failed ({rc}): {load().omfs_last_error().decode()}")


def ptr(t) -> int:
    """Device pointer of a torch tensor (None -> NULL)."""
    return 0 if t is None else t.data_ptr()


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream

"""Training targets and `gt/` images: what the absent upstream loader did with a dataset frame before the loss saw it
(the reference only passes the switches through: `--resolution` and `--white_background`,
`02_Visual_Engine/train_ghost.py:227-240`; the evaluator reads `renders/` against `gt/`,
`validation_reporting.py:60-78`).  One routine serves engine/train.py and engine/render.py, so the two agree by
construction: the frame's image (and its matte: `fg_mask_path`, or the image's alpha channel) is uploaded as decoded,
resized to the training resolution by PIL's BOX rule and composited on the run's background ON THE DEVICE
(`omfs_prepare_target`)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from .. import _lib as L
from . import io_formats as IO


def training_size(cam: dict, resolution: int) -> tuple:
    """(width, height) a view is trained at: --resolution 1/2/4/8 divides, a larger value is the target width (upstream's
    rule), anything else keeps the dataset's size."""
    if resolution in (1, 2, 4, 8):
        return cam["width"] // resolution, cam["height"] // resolution
    if resolution > 8:
        return int(resolution), int(round(cam["height"] * resolution / cam["width"]))
    return cam["width"], cam["height"]


def scaled_camera(cam: dict, w: int, h: int) -> dict:
    if (w, h) == (cam["width"], cam["height"]):
        return cam
    s = w / cam["width"]
    return {**cam, "width": w, "height": h, "fl_x": cam["fl_x"] * s, "fl_y": cam["fl_y"] * s}


def load_frame_pixels(source_path: str, frame: dict):
    """(rgb or rgba uint8 [H][W][C], separate matte uint8 [H][W] or None) of a dataset frame, as decoded."""
    rgb, alpha = IO.load_image_rgba(os.path.join(source_path, frame["file_path"]))
    mask_rel = frame.get("fg_mask_path")
    mask, mask_path = None, None
    if mask_rel:
        # render_surgery's edited copy of a dataset links images/ only (`render_surgery.py:157-163`): the mattes are then found
        # next to the real images directory
        real_root = os.path.dirname(os.path.dirname(os.path.realpath(os.path.join(source_path, frame["file_path"]))))
        for root in (source_path, real_root):
            if os.path.exists(os.path.join(root, mask_rel)):
                mask_path = os.path.join(root, mask_rel)
                break
    if mask_path:
        mask = np.ascontiguousarray(IO.read_png(mask_path)[:, :, 0])
    elif alpha is not None:                 # the matte travels as the image's alpha channel (upstream-style datasets)
        mask = np.ascontiguousarray(alpha)
    return np.ascontiguousarray(rgb), mask


def prepare_target(rgb: np.ndarray, mask, w: int, h: int, bg, as_u8: bool = False, device="cuda") -> torch.Tensor:
    """[3][h][w] fp32 in [0,1] (or, as_u8, [h][w][3] uint8) on the device: resized, matted on `bg`."""
    if rgb.dtype != np.uint8 or rgb.ndim != 3 or rgb.shape[2] not in (1, 3, 4):
        raise ValueError("rgb must be uint8 [H][W][1|3|4]")
    sh, sw, ch = rgb.shape
    if mask is not None and (mask.dtype != np.uint8 or mask.shape != (sh, sw)):
        raise ValueError("the matte must be uint8 with the image's height and width")
    src = torch.from_numpy(np.ascontiguousarray(rgb)).to(device)
    m = torch.from_numpy(np.ascontiguousarray(mask)).to(device) if mask is not None else None
    out = torch.empty((h, w, 3), dtype=torch.uint8, device=device) if as_u8 else torch.empty((3, h, w), dtype=torch.float32, device=device)
    bgc = (C.c_float * 3)(*[float(x) for x in bg])
    L.check(L.load().omfs_prepare_target(L.ptr(src), ch, sw, sh, L.ptr(m), w, h, bgc, 0 if as_u8 else L.ptr(out),
                                         L.ptr(out) if as_u8 else 0, L.stream_ptr()), "omfs_prepare_target")
    torch.cuda.current_stream().synchronize()     # src / m are released on return
    return out

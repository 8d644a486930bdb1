"""File formats on the engine's boundary: PNG (8-bit, zlib only -- no imaging dependency on the
GPU box), the Gaussian point-cloud PLY of `<model>/point_cloud/iteration_N/`, and the dataset
directory layout the reference's converter writes (`02_Visual_Engine/preprocess_video.py:200-426`)
and its launchers validate (`train_ghost.py:46-65`, `render_surgery.py:144-242`).
"""
from __future__ import annotations

import json
import math
import os
import struct
import zlib
from pathlib import Path

import numpy as np

# ------------------------------------------------------------------ PNG
_PNG_SIG = b"\x89PNG\r\n\x1a\n"


def _chunk(tag: bytes, data: bytes) -> bytes:
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def encode_png(img: np.ndarray, level: int = 1) -> bytes:
    """uint8 (H,W,3|4|1) or (H,W) -> PNG bytes (filter 0 rows, zlib `level`)."""
    a = np.ascontiguousarray(img, np.uint8)
    if a.ndim == 2:
        a = a[:, :, None]
    h, w, c = a.shape
    ctype = {1: 0, 3: 2, 4: 6}[c]
    raw = np.empty((h, 1 + w * c), np.uint8)
    raw[:, 0] = 0
    raw[:, 1:] = a.reshape(h, w * c)
    # Z_RLE: run-length matches only -- on rendered frames the same size as level-1 deflate at a third of the time
    # (28 ms vs 79 ms for 1080p on one core); the egress of render_surgery is bound by this encode
    co = zlib.compressobj(level, zlib.DEFLATED, 15, 8, zlib.Z_RLE)
    return _PNG_SIG + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0)) + \
        _chunk(b"IDAT", co.compress(raw.tobytes()) + co.flush()) + _chunk(b"IEND", b"")


def encode_png_rows(rows, width: int, height: int, level: int = 1) -> bytes:
    """PNG from ready-made RGB scanlines (`Rasterizer.to_png_rows`: [H][1 + 3W] uint8, filter byte 0 first): the buffer is
    deflated as it is -- no copy, and zlib runs without the interpreter lock."""
    a = np.asarray(rows)
    if a.dtype != np.uint8 or a.shape != (height, 1 + 3 * width) or not a.flags.c_contiguous:
        raise ValueError("rows must be a contiguous uint8 array [height][1 + 3*width]")
    co = zlib.compressobj(level, zlib.DEFLATED, 15, 8, zlib.Z_RLE)
    return _PNG_SIG + _chunk(b"IHDR", struct.pack(">IIBBBBB", width, height, 8, 2, 0, 0, 0)) + \
        _chunk(b"IDAT", co.compress(memoryview(a).cast("B")) + co.flush()) + _chunk(b"IEND", b"")


def png_parts_from_zlib_stream(stream, width: int, height: int) -> list:
    """The pieces of a PNG file around a ready-made zlib stream of the RGB scanlines (`Rasterizer.to_png_stream`: deflated on
    the device): [signature + IHDR + IDAT header, the stream itself (not copied), IDAT CRC-32 + IEND].  Chunk framing and the
    CRC are all that is left for the host; `f.writelines(parts)` writes the file without building it in memory first."""
    mv = memoryview(stream).cast("B") if not isinstance(stream, (bytes, bytearray)) else stream
    head = _PNG_SIG + _chunk(b"IHDR", struct.pack(">IIBBBBB", width, height, 8, 2, 0, 0, 0)) + struct.pack(">I", len(mv)) + b"IDAT"
    crc = zlib.crc32(mv, zlib.crc32(b"IDAT")) & 0xFFFFFFFF          # releases the interpreter lock on large buffers
    return [head, mv, struct.pack(">I", crc) + _chunk(b"IEND", b"")]


def png_from_zlib_stream(stream, width: int, height: int) -> bytes:
    return b"".join(png_parts_from_zlib_stream(stream, width, height))


def write_png(path, img: np.ndarray, level: int = 1) -> None:
    with open(path, "wb") as f:
        f.write(encode_png(img, level))


def read_png(path) -> np.ndarray:
    """8-bit non-interlaced gray / RGB / RGBA / palette PNG -> uint8 (H,W,C)."""
    data = Path(path).read_bytes()
    if data[:8] != _PNG_SIG:
        raise ValueError(f"{path}: not a PNG file")
    pos, idat, palette = 8, [], None
    w = h = depth = ctype = interlace = None
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        pos += 12 + n
        if tag == b"IHDR":
            w, h, depth, ctype, _, _, interlace = struct.unpack(">IIBBBBB", body)
        elif tag == b"PLTE":
            palette = np.frombuffer(body, np.uint8).reshape(-1, 3)
        elif tag == b"IDAT":
            idat.append(body)
        elif tag == b"IEND":
            break
    if depth != 8 or interlace != 0 or ctype not in (0, 2, 3, 4, 6):
        raise ValueError(f"{path}: only 8-bit non-interlaced PNGs are supported (depth {depth}, type {ctype}, interlace {interlace})")
    c = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    stride = w * c
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), np.uint8).reshape(h, stride + 1)
    out = np.zeros((h, stride), np.uint8)
    prev = np.zeros(stride, np.int32)
    for y in range(h):
        ft, line = int(raw[y, 0]), raw[y, 1:].astype(np.int32)
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = (line + prev) & 255
        elif ft == 1:
            cur = line.copy()
            for i in range(c, stride, c):
                cur[i:i + c] = (cur[i:i + c] + cur[i - c:i]) & 255
        elif ft == 3:
            cur = line.copy()
            for i in range(0, stride, c):
                left = cur[i - c:i] if i >= c else 0
                cur[i:i + c] = (cur[i:i + c] + ((left + prev[i:i + c]) >> 1)) & 255
        elif ft == 4:
            cur = line.copy()
            for i in range(0, stride, c):
                a = cur[i - c:i] if i >= c else np.zeros(c, np.int32)
                b = prev[i:i + c]
                cc = prev[i - c:i] if i >= c else np.zeros(c, np.int32)
                p = a + b - cc
                pa, pb, pc = np.abs(p - a), np.abs(p - b), np.abs(p - cc)
                pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, b, cc))
                cur[i:i + c] = (cur[i:i + c] + pred) & 255
        else:
            raise ValueError(f"{path}: bad PNG filter {ft}")
        out[y] = cur
        prev = cur
    img = out.reshape(h, w, c)
    if ctype == 3:
        img = palette[img[:, :, 0]]
    elif ctype == 4:
        img = np.concatenate([np.repeat(img[:, :, :1], 3, 2), img[:, :, 1:]], 2)
    return img


def load_image_rgba(path):
    """((H,W,3) uint8, alpha (H,W) uint8 or None): training images of upstream-style datasets may carry the foreground
    matte as an alpha channel instead of a separate `fg_mask_path`."""
    try:
        from PIL import Image
        with Image.open(path) as im:
            if im.mode in ("RGBA", "LA") or (im.mode == "P" and "transparency" in im.info):
                a = np.asarray(im.convert("RGBA"))
                return np.ascontiguousarray(a[:, :, :3]), np.ascontiguousarray(a[:, :, 3])
            return np.asarray(im.convert("RGB")), None
    except ImportError:
        pass
    img = read_png(path)
    if img.shape[2] in (2, 4):
        alpha = np.ascontiguousarray(img[:, :, -1])
        rgb = np.repeat(img[:, :, :1], 3, 2) if img.shape[2] == 2 else np.ascontiguousarray(img[:, :, :3])
        return rgb, alpha
    return (np.repeat(img, 3, 2) if img.shape[2] == 1 else img[:, :, :3]), None


def load_image_rgb(path) -> np.ndarray:
    """(H,W,3) uint8; an alpha channel, if any, is returned separately by load_image_rgba.  Decoded by PIL when it is
    installed (the reference's own dependency, `validation_reporting.py:11`; ~20x faster than the numpy decoder below on
    adaptively filtered 1080p PNGs), by `read_png` otherwise."""
    try:
        from PIL import Image
        with Image.open(path) as im:
            return np.asarray(im.convert("RGB"))
    except ImportError:
        pass
    img = read_png(path)
    if img.shape[2] == 1:
        img = np.repeat(img, 3, 2)
    return img[:, :, :3]


# ------------------------------------------------------------------ PLY (Gaussian point cloud)
def _ply_fields(with_binding: bool):
    names = ["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(3)] + [f"f_rest_{i}" for i in range(45)] + \
        ["opacity"] + [f"scale_{i}" for i in range(3)] + [f"rot_{i}" for i in range(4)]
    if with_binding:
        names.append("binding_0")
    return names


def save_gaussian_ply(path, g: dict) -> None:
    """3DGS point_cloud.ply attribute order (SH rest stored channel-major) + `binding_0`."""
    n = g["xyz"].shape[0]
    sh = np.asarray(g["sh"], np.float32)                           # (N,16,3)
    cols = [np.asarray(g["xyz"], np.float32), np.zeros((n, 3), np.float32), sh[:, 0, :],
            sh[:, 1:, :].transpose(0, 2, 1).reshape(n, 45), np.asarray(g["opacity"], np.float32).reshape(n, 1),
            np.asarray(g["log_scale"], np.float32), np.asarray(g["rot"], np.float32),
            np.asarray(g["binding"], np.float32).reshape(n, 1)]
    table = np.concatenate(cols, 1).astype("<f4")
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % n + \
        "".join(f"property float {name}\n" for name in _ply_fields(True)) + "end_header\n"
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(table.tobytes())


def load_gaussian_ply(path) -> dict:
    with open(path, "rb") as f:
        blob = f.read()
    end = blob.index(b"end_header\n") + len(b"end_header\n")
    lines = blob[:end].decode("ascii").splitlines()
    n = next(int(l.split()[-1]) for l in lines if l.startswith("element vertex"))
    names = [l.split()[-1] for l in lines if l.startswith("property float")]
    table = np.frombuffer(blob[end:end + 4 * n * len(names)], "<f4").reshape(n, len(names))
    col = {name: table[:, i] for i, name in enumerate(names)}
    sh = np.empty((n, 16, 3), np.float32)
    sh[:, 0, :] = np.stack([col[f"f_dc_{i}"] for i in range(3)], 1)
    sh[:, 1:, :] = np.stack([col[f"f_rest_{i}"] for i in range(45)], 1).reshape(n, 3, 15).transpose(0, 2, 1)
    return {"xyz": np.stack([col["x"], col["y"], col["z"]], 1).copy(), "log_scale": np.stack([col[f"scale_{i}"] for i in range(3)], 1).copy(),
            "rot": np.stack([col[f"rot_{i}"] for i in range(4)], 1).copy(), "opacity": col["opacity"].copy(), "sh": sh,
            "binding": (col["binding_0"] if "binding_0" in col else np.zeros(n)).astype(np.int32)}


# ------------------------------------------------------------------ dataset directory
FLAME_KEYS = ("shape", "expr", "rotation", "neck_pose", "jaw_pose", "eyes_pose", "translation", "static_offset", "dynamic_offset")


def camera_from_frame(frame: dict, top: dict) -> dict:
    """transforms frame (NeRF/OpenGL camera-to-world, `preprocess_video.py:372-401`) -> engine camera dict
    (y-down, z-forward world-to-view; focal from camera_angle_x = 2 atan(w / (2 fl_x)), `:237`)."""
    w = int(frame.get("w", top.get("w")))
    h = int(frame.get("h", top.get("h")))
    c2w = np.array(frame["transform_matrix"], np.float64)
    c2w[:3, 1:3] *= -1.0
    w2v = np.linalg.inv(c2w)
    if "fl_x" in frame:
        fl_x = float(frame["fl_x"])
    elif "camera_angle_x" in frame or "camera_angle_x" in top:
        fl_x = w / (2.0 * math.tan(float(frame.get("camera_angle_x", top.get("camera_angle_x"))) / 2.0))
    else:
        fl_x = float(top["fl_x"])
    fl_y = float(frame.get("fl_y", fl_x))
    return {"width": w, "height": h, "fl_x": fl_x, "fl_y": fl_y, "tanfovx": w / (2.0 * fl_x), "tanfovy": h / (2.0 * fl_y),
            "world_to_view": w2v.astype(np.float32), "cam_pos": c2w[:3, 3].astype(np.float32),
            "camera_angle_x": 2.0 * math.atan(w / (2.0 * fl_x))}


def frame_from_camera(cam: dict) -> list:
    """Inverse of camera_from_frame: 4x4 OpenGL camera-to-world as nested lists."""
    w2v = np.asarray(cam["world_to_view"], np.float64)
    c2w = np.linalg.inv(w2v)
    c2w[:3, 1:3] *= -1.0
    return c2w.tolist()


def _flame_row(d: dict, key: str, width: int) -> np.ndarray:
    """First row of a per-frame FLAME array stored as (width,) or (1,width); zeros if the key is absent."""
    if key not in d:
        return np.zeros(width, np.float32)
    return np.asarray(d[key], np.float32).reshape(-1, width)[0]


def load_split(data_dir: str, split: str = "train") -> dict:
    """Read transforms_<split>.json and the FLAME parameters of every frame (per-frame
    `flame_param_path` files win over the batched flame_param.npz, which is how
    `render_surgery.py:203-218` routes the edited parameters to the renderer).
    Returns {"frames", "top", "flame": batched dict over the split's distinct timesteps,
    "timestep_of_frame": row of `flame` used by each frame}."""
    root = Path(data_dir)
    with open(root / f"transforms_{split}.json", "r") as f:
        top = json.load(f)
    frames = top.get("frames", [])
    batched = dict(np.load(root / "flame_param.npz", allow_pickle=True)) if (root / "flame_param.npz").exists() else None
    n_batched = np.asarray(batched["expr"]).reshape(-1, np.asarray(batched["expr"]).shape[-1]).shape[0] if batched is not None else 0
    per_t: dict = {}
    for i, fr in enumerate(frames):
        t = int(fr.get("timestep_index", i))
        if t in per_t:
            continue
        rel = fr.get("flame_param_path")
        if rel and (root / rel).exists():
            per_t[t] = dict(np.load(root / rel, allow_pickle=True))
        elif batched is not None:
            if t >= n_batched:
                raise IndexError(f"timestep {t} outside flame_param.npz ({n_batched} frames)")
            per_t[t] = {k: (np.asarray(v)[t:t + 1] if (np.asarray(v).ndim >= 2 and np.asarray(v).shape[0] == n_batched and k != "static_offset")
                            else np.asarray(v)) for k, v in batched.items()}
        else:
            raise FileNotFoundError(f"no FLAME parameters for timestep {t} in {data_dir}")
    if not per_t:
        raise ValueError(f"transforms_{split}.json holds no frames")
    order = sorted(per_t)
    index_of = {t: i for i, t in enumerate(order)}
    first = per_t[order[0]]
    n_expr = int(np.asarray(first["expr"]).shape[-1])
    widths = {"expr": n_expr, "rotation": 3, "neck_pose": 3, "jaw_pose": 3, "eyes_pose": 6, "translation": 3}
    flame = {k: np.stack([_flame_row(per_t[t], k, w) for t in order]) for k, w in widths.items()}
    flame["shape"] = np.asarray(first["shape"], np.float32).reshape(-1)
    if "static_offset" in first:
        flame["static_offset"] = np.asarray(first["static_offset"], np.float32)
    if any("dynamic_offset" in per_t[t] and np.any(per_t[t]["dynamic_offset"]) for t in order):
        n_v = np.asarray(first["dynamic_offset"]).shape[-2]
        flame["dynamic_offset"] = np.stack([np.asarray(per_t[t]["dynamic_offset"], np.float32).reshape(-1, n_v, 3)[0] for t in order])
    return {"frames": frames, "top": top, "flame": flame,
            "timestep_of_frame": [index_of[int(fr.get("timestep_index", i))] for i, fr in enumerate(frames)]}


def write_dataset(root, cams: list, timesteps: list, images: list, flame: dict, fg_masks: bool = False) -> None:
    """Write a dataset in the converter's layout (`preprocess_video.py:314-416`): images/%05d_%02d.png,
    flame_param/%05d.npz, flame_param.npz, canonical_flame_param.npz, transforms_{train,test,val}.json
    (90/10 split, val = test) and transforms.json.  `images[i]` is uint8 (H,W,3) for view i."""
    root = Path(root)
    (root / "images").mkdir(parents=True, exist_ok=True)
    (root / "flame_param").mkdir(exist_ok=True)
    T = np.asarray(flame["expr"]).shape[0]
    np.savez(root / "flame_param.npz", **{k: np.asarray(v) for k, v in flame.items()})
    for t in range(T):
        per = {k: (np.asarray(v)[t:t + 1] if (np.asarray(v).ndim >= 2 and np.asarray(v).shape[0] == T and k != "static_offset") else np.asarray(v))
               for k, v in flame.items()}
        np.savez(root / "flame_param" / f"{t:05d}.npz", **per)
    canon = {k: (np.zeros((1,) + np.asarray(v).shape[1:], np.float32) if (np.asarray(v).ndim >= 2 and np.asarray(v).shape[0] == T and k != "static_offset")
                 else np.asarray(v)) for k, v in flame.items()}
    np.savez(root / "canonical_flame_param.npz", **canon)
    if fg_masks:
        (root / "fg_masks").mkdir(exist_ok=True)
    entries = []
    for i, (cam, t, img) in enumerate(zip(cams, timesteps, images)):
        name = f"{t:05d}_{i % 100:02d}.png" if len(cams) != T else f"{t:05d}_00.png"
        write_png(root / "images" / name, img)
        e = {"file_path": f"images/{name}", "flame_param_path": f"flame_param/{t:05d}.npz", "transform_matrix": frame_from_camera(cam),
             "timestep_index": int(t), "camera_index": 0, "camera_angle_x": cam["camera_angle_x"], "w": cam["width"], "h": cam["height"]}
        if fg_masks:
            write_png(root / "fg_masks" / name, np.full(img.shape[:2], 255, np.uint8))
            e["fg_mask_path"] = f"fg_masks/{name}"
        entries.append(e)
    c0 = cams[0]
    top = {"camera_angle_x": c0["camera_angle_x"], "camera_angle_y": 2.0 * math.atan(c0["height"] / (2.0 * c0["fl_y"])),
           "fl_x": c0["fl_x"], "fl_y": c0["fl_y"], "cx": c0["width"] / 2.0, "cy": c0["height"] / 2.0, "w": c0["width"], "h": c0["height"],
           "timestep_indices": list(range(T)), "camera_indices": [0]}
    n = len(entries)
    split = max(1, n - n // 10)
    for name, fr in (("transforms_train.json", entries[:split]), ("transforms_test.json", entries[split:]),
                     ("transforms_val.json", entries[split:]), ("transforms.json", entries)):
        with open(root / name, "w") as f:
            json.dump({**top, "frames": fr}, f, indent=2)

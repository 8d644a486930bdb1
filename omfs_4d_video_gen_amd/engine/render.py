#!/usr/bin/env python3
"""Engine entry point with the CLI of the absent upstream `gaussian_avatars_repo/render.py`, i.e. the
argv `02_Visual_Engine/render_surgery.py:289-301` emits:

  render.py --source_path D --model_path M --bind_to_mesh --skip_val --skip_test [--iteration K]

Renders the TRAIN split of D (whose FLAME parameters render_surgery has edited) with the Gaussians
of M/point_cloud/iteration_K and writes M/train/ours_K/renders/%05d.png and .../gt/%05d.png
(`render_surgery.py:324-362`, `validation_reporting.py:60-78`).  With WORLD_SIZE>1 frames shard
across ranks (frame f -> rank f mod world); no collective is needed.
"""
from __future__ import annotations

import argparse
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

_PKG_ROOT = Path(__file__).resolve().parents[2]
if str(_PKG_ROOT) not in sys.path:
    sys.path.insert(0, str(_PKG_ROOT))

import torch  # noqa: E402


def parse(argv=None):
    p = argparse.ArgumentParser(description="Render a FLAME-rigged Gaussian avatar (MI355X engine).")
    p.add_argument("--source_path", "-s", required=True)
    p.add_argument("--model_path", "-m", required=True)
    p.add_argument("--bind_to_mesh", action="store_true")
    p.add_argument("--iteration", type=int, default=-1)
    p.add_argument("--skip_train", action="store_true")
    p.add_argument("--skip_val", action="store_true")
    p.add_argument("--skip_test", action="store_true")
    p.add_argument("--white_background", action="store_true")
    p.add_argument("--resolution", "-r", type=int, default=None, help="default: the resolution the model was trained at (cfg_args.json)")
    p.add_argument("--sh_degree", type=int, default=3)
    p.add_argument("--png_workers", type=int, default=8)
    p.add_argument("--dup_capacity", type=int, default=0, help="initial (Gaussian, tile) pair capacity (0 = sized from the cloud and the image; grown, and the split rendered again, on overflow)")
    args, unknown = p.parse_known_args(argv)
    if unknown:
        print(f"[engine] ignoring unknown arguments: {unknown}")
    return args


def find_iteration(model_path: str, wanted: int) -> int:
    pc = Path(model_path) / "point_cloud"
    have = sorted(int(d.name.split("_")[1]) for d in pc.iterdir() if d.name.startswith("iteration_")) if pc.is_dir() else []
    if not have:
        raise FileNotFoundError(f"no point_cloud/iteration_* under {model_path}")
    if wanted > 0:
        if wanted not in have:
            raise FileNotFoundError(f"iteration {wanted} not found under {pc} (have {have})")
        return wanted
    return have[-1]


TUNED_KEYS = ("expr", "rotation", "neck_pose", "jaw_pose", "eyes_pose", "translation")


def tuned_flame(pc_dir: Path, dataset_flame: dict) -> dict:
    """FLAME sequence to render with.  A model trained with --finetune_flame_params stores the tuned sequence and the
    sequence it started from; the dataset being rendered may carry edits (render_surgery rewrites translation / jaw):
    render `tuned + (dataset - source)`.  Without tuned parameters, or when the sequences do not match, the dataset's."""
    import numpy as np
    src_p, tuned_p = pc_dir / "flame_param_source.npz", pc_dir / "flame_param.npz"
    if not (src_p.exists() and tuned_p.exists()):
        return dataset_flame
    src, tuned = dict(np.load(src_p)), dict(np.load(tuned_p))
    out = dict(dataset_flame)
    for k in TUNED_KEYS:
        if k not in dataset_flame or k not in src or k not in tuned:
            continue
        d, s0, t = (np.asarray(x, np.float32) for x in (dataset_flame[k], src[k], tuned[k]))
        if s0.shape != t.shape or d.size != s0.size:
            return dataset_flame
        out[k] = (t + (d.reshape(s0.shape) - s0)).astype(np.float32)
    return out


def render_split(args, split_name: str, it: int, rank: int, world: int):
    import json
    from omfs_4d_video_gen_amd.engine import io_formats as IO
    from omfs_4d_video_gen_amd.engine.rig_loader import load_rig
    from omfs_4d_video_gen_amd.engine.trainer import Renderer, View
    split = IO.load_split(args.source_path, split_name)
    g = IO.load_gaussian_ply(Path(args.model_path) / "point_cloud" / f"iteration_{it}" / "point_cloud.ply")
    from omfs_4d_video_gen_amd.engine import targets as TG
    cfg_p = Path(args.model_path) / "cfg_args.json"
    cfg = json.load(open(cfg_p)) if cfg_p.exists() else {}
    # as upstream merges the model's cfg_args into the render arguments: background and resolution are the training run's
    white = args.white_background or bool(cfg.get("white_background", False))
    resolution = args.resolution if args.resolution is not None else int(cfg.get("resolution", -1))
    bg = (1.0, 1.0, 1.0) if white else (0.0, 0.0, 0.0)
    cams = [IO.camera_from_frame(fr, split["top"]) for fr in split["frames"]]
    if not cams:
        return 0
    cams = [TG.scaled_camera(c, *TG.training_size(c, resolution)) for c in cams]
    w, h = cams[0]["width"], cams[0]["height"]
    out_dir = Path(args.model_path) / split_name / f"ours_{it}"
    (out_dir / "renders").mkdir(parents=True, exist_ok=True)
    (out_dir / "gt").mkdir(parents=True, exist_ok=True)
    flame = tuned_flame(Path(args.model_path) / "point_cloud" / f"iteration_{it}", split["flame"])
    # the frames of a sequence are independent: three streams with raster buffers of their own (see Renderer)
    r = Renderer(load_rig(), flame, g, w, h, bg=bg, sh_degree=args.sh_degree, n_streams=int(os.environ.get("OMFS_RENDER_STREAMS", "3")),
                 dup_capacity=args.dup_capacity or None)
    pool = ThreadPoolExecutor(max_workers=max(1, args.png_workers))
    pending, pending_gt = [], []
    mine = range(rank, len(cams), world)
    n_slots = max(4, 2 * args.png_workers)

    def encode_and_write(path, k, event):
        # the frame was deflated on the device (omfs_png_deflate): fetch the zlib stream, add the PNG framing + chunk CRC, write
        with open(path, "wb") as f:
            f.writelines(IO.png_parts_from_zlib_stream(r.fetch_png_stream(k, event), w, h))

    while True:
        for idx in mine:
            view = View(cams[idx], split["timestep_of_frame"][idx])
            if len(pending) >= n_slots:                           # the pinned ring is reused: keep at most n_slots frames in flight
                pending.pop(0).result()
            k, event = r.render_png_stream(view, n_slots)         # GPU-side scanlines + deflate into ring slot k
            pending.append(pool.submit(encode_and_write, out_dir / "renders" / f"{idx:05d}.png", k, event))
            # gt/: the target the trainer was shown for this frame -- resized to the training resolution, matted on the run's
            # background (engine/targets.py) -- so an evaluator compares like with like (validation_reporting.py:60-78)
            if os.path.exists(os.path.join(args.source_path, split["frames"][idx]["file_path"])):
                rgb, mask = TG.load_frame_pixels(args.source_path, split["frames"][idx])
                gt = TG.prepare_target(rgb, mask, w, h, bg, as_u8=True).cpu().numpy()
                pending_gt.append(pool.submit(IO.write_png, out_dir / "gt" / f"{idx:05d}.png", gt))
        for f in pending + pending_gt:
            f.result()
        pending, pending_gt = [], []
        if not r.overflowed():
            break
        # the default pair capacity is sized for a typical head (20 pairs per Gaussian at 1080p); a scene beyond it rendered empty
        # lists from the overflowing frame on: grow the buffers and render this rank's frames of the split again
        print(f"[engine] tile-list capacity exceeded: {r.grow_dup_capacity(2.0)} pairs now, rendering the split again", flush=True)
    pool.shutdown()
    r.check_status()
    return len(mine)


def main(argv=None):
    args = parse(argv)
    if not args.bind_to_mesh:
        raise SystemExit("[engine] only --bind_to_mesh (FLAME-rigged) rendering is implemented")
    if not torch.cuda.is_available():
        raise SystemExit("[engine] no GPU visible: the engine has no CPU path")
    world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("OMFS_DIST_BACKEND", "nccl")   # gloo: functional rehearsal with ranks sharing a card
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    it = find_iteration(args.model_path, args.iteration)
    t0 = time.time()
    total = 0
    for name, skip in (("train", args.skip_train), ("val", args.skip_val), ("test", args.skip_test)):
        if skip or not os.path.exists(os.path.join(args.source_path, f"transforms_{name}.json")):
            continue
        n = render_split(args, name, it, rank, world)
        total += n
        print(f"[engine] rendered {n} {name} frames at iteration {it} (rank {rank}/{world})", flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(f"Rendering complete: {total} frames on rank 0 in {time.time() - t0:.1f} s")


if __name__ == "__main__":
    main()

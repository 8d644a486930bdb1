#!/usr/bin/env python3
"""Engine entry point with the CLI of the absent upstream `gaussian_avatars_repo/train.py`, i.e. the
argv `02_Visual_Engine/train_ghost.py:227-240` emits:

  train.py --source_path D --model_path M --bind_to_mesh --iterations N --resolution R
           --save_iterations ... --checkpoint_iterations ... [--white_background]

Outputs (what `render_surgery.py:271-287` and `train_ghost.py:141-156` look for):
  M/point_cloud/iteration_<it>/point_cloud.ply (+ flame_param.npz), M/chkpnt<it>.pth, M/cfg_args.json.
Progress lines contain "iteration <n>" for the UI regex (`app.py:1387-1398`).
Multi-GPU: launch with torch.distributed.run; views shard across ranks, gradients are all-reduced.
The per-timestep FLAME expression / poses / translation are optimised with the Gaussians BY DEFAULT, as upstream does for
--bind_to_mesh (the reference's argv passes no opt-out, `train_ghost.py:227-237`); `--not_finetune_flame_params` keeps the
sequence fixed (`--finetune_flame_params` is accepted for older command lines): the tuned sequence is saved as point_cloud/iteration_<it>/flame_param.npz next to flame_param_source.npz (as loaded), and
render.py applies "tuned + (dataset - source)" so that render_surgery's edits still act on the tuned sequence.
`--start_checkpoint M/chkpnt<it>.pth` resumes (parameters, Adam moments, SH degree, FLAME state, iteration).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

_PKG_ROOT = Path(__file__).resolve().parents[2]
if str(_PKG_ROOT) not in sys.path:
    sys.path.insert(0, str(_PKG_ROOT))

import numpy as np  # noqa: E402
import torch  # noqa: E402


def parse(argv=None):
    p = argparse.ArgumentParser(description="Train a FLAME-rigged Gaussian avatar (MI355X engine).")
    p.add_argument("--source_path", "-s", required=True)
    p.add_argument("--model_path", "-m", required=True)
    p.add_argument("--bind_to_mesh", action="store_true")
    p.add_argument("--iterations", type=int, default=30000)
    p.add_argument("--resolution", "-r", type=int, default=-1)
    p.add_argument("--save_iterations", nargs="+", type=int, default=[])
    p.add_argument("--checkpoint_iterations", nargs="+", type=int, default=[])
    p.add_argument("--white_background", action="store_true")
    p.add_argument("--sh_degree", type=int, default=3)
    p.add_argument("--n_gaussians", type=int, default=0, help="fixed Gaussian count (0 = 10 per triangle)")
    p.add_argument("--seed", type=int, default=0)
    # adaptive density control, GaussianAvatars' schedule
    p.add_argument("--densify_from_iter", type=int, default=10000)
    p.add_argument("--densify_until_iter", type=int, default=600000)
    p.add_argument("--densification_interval", type=int, default=2000)
    p.add_argument("--densify_grad_threshold", type=float, default=2e-4)
    p.add_argument("--opacity_reset_interval", type=int, default=60000)
    p.add_argument("--max_gaussians", type=int, default=0, help="capacity for densification (0 = 4x the initial count)")
    p.add_argument("--no_densify", action="store_true")
    p.add_argument("--log_every", type=int, default=100)
    p.add_argument("--finetune_flame_params", action="store_true", help="the default (accepted for older command lines)")
    p.add_argument("--not_finetune_flame_params", action="store_true", help="keep the FLAME sequence fixed (upstream's opt-out)")
    p.add_argument("--flame_expr_lr", type=float, default=1e-3)
    p.add_argument("--flame_pose_lr", type=float, default=1e-5)
    p.add_argument("--flame_trans_lr", type=float, default=1e-6)
    p.add_argument("--start_checkpoint", type=str, default=None)
    p.add_argument("--dup_capacity", type=int, default=0, help="initial (Gaussian, tile) pair capacity (0 = sized from the cloud and the image; grown on overflow)")
    p.add_argument("--coherent_order", action="store_true",
                   help="store the cloud along a Morton curve over its parent triangles (measured neutral on MI355X: projection kernels "
                        "gain what the atomic-based binning loses; off by default)")
    p.add_argument("--no_shuffle", action="store_true", help="visit the views in index order instead of a seeded random order per epoch")
    p.add_argument("--deterministic", action="store_true",
                   help="bit-reproducible training (= OMFS_DETERMINISTIC=1): fixed-point gradient accumulation instead of float atomics, ~4 %% slower")
    p.add_argument("--target_storage", choices=("auto", "f32", "u8"), default="auto",
                   help="how training images are kept in HBM: fp32 planes, 8-bit RGB expanded per step, or by dataset size")
    args, unknown = p.parse_known_args(argv)
    if unknown:
        print(f"[engine] ignoring unknown arguments: {unknown}")
    if args.finetune_flame_params and args.not_finetune_flame_params:
        p.error("--finetune_flame_params and --not_finetune_flame_params exclude each other")
    args.finetune_flame_params = not args.not_finetune_flame_params
    return args


def initial_gaussians(n: int, n_faces: int, seed: int) -> dict:
    """Mesh-bound start: k Gaussians per triangle spread in its plane, triangle-sized, grey, faint."""
    rng = np.random.default_rng(seed)
    xyz = np.zeros((n, 3), np.float32)
    xyz[:, 0] = rng.standard_normal(n) * 0.2
    xyz[:, 2] = rng.standard_normal(n) * 0.2
    sh = np.zeros((n, 16, 3), np.float32)
    rot = np.zeros((n, 4), np.float32)
    rot[:, 0] = 1.0
    return {"xyz": xyz, "log_scale": np.full((n, 3), np.log(0.5), np.float32), "rot": rot,
            "opacity": np.full(n, float(np.log(0.1 / 0.9)), np.float32), "sh": sh,
            "binding": (np.arange(n) % n_faces).astype(np.int32)}


class Rollback:
    """Device-side snapshot of everything a training iteration changes (Gaussian parameters, their Adam moments, the
    densification statistics, the tuned FLAME tensors and their moments, the counters), taken at every log interval that found
    the tile-list capacity sufficient.  An interval that overflowed rendered EMPTY lists from the overflowing iteration on (the
    scan zeroes them): its Adam steps saw momentum and regularisers only.  It is not kept: the trainer is put back to the
    snapshot, the capacity grown and the interval redone (the view order is a function of the iteration, so the redo sees the
    same views).  Cost: three [59][n_pad] device copies per log interval."""

    def __init__(self, trainer):
        self.t = trainer
        self.s = None

    def take(self, it: int) -> None:
        t = self.t
        m, ft = t.model, t.flame_ft
        s = self.s
        # binding is [n], the planes are [59][n_pad] with n_pad = n rounded up to 256: a densification can change n and keep
        # n_pad (100100 -> 100200 Gaussians are both 100352 columns), so the snapshot is keyed on BOTH shapes
        if s is None or s["params"].shape != m.params.shape or s["binding"].shape != m.binding.shape:
            s = self.s = {"params": torch.empty_like(m.params), "m": torch.empty_like(t.opt.m), "v": torch.empty_like(t.opt.v),
                          "binding": torch.empty_like(m.binding)}
        s["params"].copy_(m.params); s["m"].copy_(t.opt.m); s["v"].copy_(t.opt.v); s["binding"].copy_(m.binding)
        s["stats"] = None if t.densify_stats is None else t.densify_stats.clone()
        s["flame"] = None if ft is None else ({k: v.clone() for k, v in ft.params.items()}, {k: v.clone() for k, v in ft.m.items()},
                                              {k: v.clone() for k, v in ft.v.items()}, ft.step_count)
        s["n"], s["order"] = m.n, m.order
        s["it"], s["opt_step"], s["step_idx"], s["sh_degree"] = it, t.opt.step_count, t.step_idx, t.sh_degree

    def restore(self) -> int:
        t, s = self.t, self.s
        m, ft = t.model, t.flame_ft
        if s["params"].shape != m.params.shape or s["binding"].shape != m.binding.shape or s["n"] != m.n:
            # a densification lies inside the interval (it may have kept n_pad and changed only n): put the old buffers back
            m.params, m.binding, m.n, m.n_pad = s["params"].clone(), s["binding"].clone(), s["n"], int(s["params"].shape[1])
            t.opt.m, t.opt.v = s["m"].clone(), s["v"].clone()
            t.alloc_grads(m.n_pad)
            t.rast.g2[m.n:].zero_()
        else:
            m.params.copy_(s["params"]); t.opt.m.copy_(s["m"]); t.opt.v.copy_(s["v"]); m.binding.copy_(s["binding"]); m.n = s["n"]
        m.order = s["order"]
        t.densify_stats = None if s["stats"] is None else s["stats"].clone()
        if ft is not None:
            p_, m_, v_, n_ = s["flame"]
            for k in ft.params:
                ft.params[k].copy_(p_[k]); ft.m[k].copy_(m_[k]); ft.v[k].copy_(v_[k])
            ft.step_count = n_
            ft.grad_flat.zero_()
            ft.refresh_rotmats()
        t.opt.step_count, t.step_idx, t.sh_degree = s["opt_step"], s["step_idx"], s["sh_degree"]
        t._prefetch, t._frames_ready, t._state_step = None, None, -1     # frames posed ahead belong to the discarded state
        t.invalidate_graphs()
        return s["it"]


def main(argv=None):
    args = parse(argv)
    if args.deterministic:
        os.environ["OMFS_DETERMINISTIC"] = "1"       # read by Rasterizer / FlameFineTuner when the trainer is built below
    from omfs_4d_video_gen_amd.engine import io_formats as IO
    from omfs_4d_video_gen_amd.engine.rig_loader import load_rig
    from omfs_4d_video_gen_amd.engine.trainer import Trainer, View

    if not args.bind_to_mesh:
        raise SystemExit("[engine] only --bind_to_mesh (FLAME-rigged) training is implemented")
    if not torch.cuda.is_available():
        raise SystemExit("[engine] no GPU visible: the engine has no CPU path")
    world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    pg = None
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("OMFS_DIST_BACKEND", "nccl")   # gloo: functional rehearsal with ranks sharing a card
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
        pg = dist.group.WORLD

    split = IO.load_split(args.source_path, "train")
    rig = load_rig()
    bg = (1.0, 1.0, 1.0) if args.white_background else (0.0, 0.0, 0.0)
    views, size = [], None
    from concurrent.futures import ThreadPoolExecutor
    # a rank trains on the views with index = rank (mod world) only (distributed.views_of_rank): it decodes and keeps just
    # those images; the other views contribute their cameras and timesteps (the compact exchange poses every rank's view)
    from omfs_4d_video_gen_amd.engine.distributed import views_of_rank
    mine = set(views_of_rank(len(split["frames"]), rank, world)) if not args.no_shuffle else set(range(len(split["frames"])))
    from omfs_4d_video_gen_amd.engine import targets as TG
    with ThreadPoolExecutor(max_workers=8) as pool:          # PNG decode releases the interpreter lock
        loaded = list(pool.map(lambda ifr: TG.load_frame_pixels(args.source_path, ifr[1]) if ifr[0] in mine else (None, None),
                               enumerate(split["frames"])))
    # fp32 targets are 4x the bytes of the images: kept while they fit comfortably beside the model, 8-bit otherwise
    cam0 = IO.camera_from_frame(split["frames"][0], split["top"])
    n_px = sum(1 for im, _ in loaded if im is not None) * int(np.prod(TG.training_size(cam0, args.resolution)))
    store_u8 = args.target_storage == "u8" or (args.target_storage == "auto" and n_px * 12 > 64 << 30)
    for fr, trow, (img, mask) in zip(split["frames"], split["timestep_of_frame"], loaded):
        cam = IO.camera_from_frame(fr, split["top"])
        w, h = TG.training_size(cam, args.resolution)
        cam = TG.scaled_camera(cam, w, h)
        if size is None:
            size = (w, h)
        elif size != (w, h):
            raise SystemExit("[engine] all training views must share one resolution")
        if img is None:                         # another rank's view: camera and timestep only
            views.append(View(cam, int(trow), target=None, name=os.path.basename(fr["file_path"])))
            continue
        # resize (PIL's BOX rule) + matte on the run's background on the device (engine/targets.py; render.py writes the very
        # same pixels as gt/): fp32 planes, or [H][W][3] bytes expanded one view at a time by omfs_rgb8_to_image -- a soft
        # matte edge is composited before the 8-bit rounding (at most half a level away from the fp32 composite)
        views.append(View(cam, int(trow), target=TG.prepare_target(img, mask, w, h, bg, as_u8=store_u8), name=os.path.basename(fr["file_path"])))
    del loaded
    n = args.n_gaussians if args.n_gaussians > 0 else 10 * rig.n_faces
    g0 = initial_gaussians(n, rig.n_faces, args.seed)
    ckpt = None
    if args.start_checkpoint:
        from omfs_4d_video_gen_amd.engine.gaussians import unpack_params
        ckpt = torch.load(args.start_checkpoint, map_location="cpu", weights_only=True)
        n = int(ckpt["binding"].shape[0])
        g0 = unpack_params(ckpt["params"][:, :n].numpy())
        g0["binding"] = ckpt["binding"].numpy()
    densify = not args.no_densify and args.iterations > args.densify_from_iter
    cap = (args.max_gaussians if args.max_gaussians > 0 else 4 * n) if densify else n
    trainer = Trainer(rig, split["flame"], g0, views, size[0], size[1], bg=bg, iterations=args.iterations,
                      sh_degree_max=args.sh_degree, start_sh_degree=0, rank=rank, world_size=world, process_group=pg,
                      n_capacity=cap, dup_capacity=args.dup_capacity or None, finetune_flame=args.finetune_flame_params, coherent_order=args.coherent_order,
                      shuffle_views=None if args.no_shuffle else args.seed,
                      flame_lr={"expr": args.flame_expr_lr, "pose": args.flame_pose_lr, "translation": args.flame_trans_lr})
    it0 = 0
    if ckpt is not None:
        it0 = int(ckpt["iteration"])
        # the checkpoint's columns are in ITS storage order; this trainer's cloud is g0[order]
        order = trainer.model.order if trainer.model.order is not None else np.arange(n)
        idx = torch.from_numpy(np.ascontiguousarray(order))
        trainer.opt.m[:, :n].copy_(ckpt["adam_m"][:, :n][:, idx])
        trainer.opt.v[:, :n].copy_(ckpt["adam_v"][:, :n][:, idx])
        trainer.opt.step_count = it0
        trainer.sh_degree = int(ckpt["sh_degree"])
        trainer.step_idx = it0
        if trainer.flame_ft is not None and "flame" in ckpt:
            trainer.flame_ft.load_state_dict(ckpt["flame"])
        if rank == 0:
            print(f"[engine] resumed from {args.start_checkpoint} at iteration {it0} with {n} Gaussians", flush=True)
    controller = None
    if densify:
        from omfs_4d_video_gen_amd.engine.densify import DensityController, scene_extent
        controller = DensityController(trainer, scene_extent(views), args.densify_from_iter, args.densify_until_iter,
                                       args.densification_interval, args.densify_grad_threshold,
                                       opacity_reset_interval=args.opacity_reset_interval, max_gaussians=cap, seed=args.seed)
        if ckpt is not None and "densify_stats" in ckpt:     # the statistics gathered since the last densification go on
            order = trainer.model.order if trainer.model.order is not None else np.arange(n)
            trainer.densify_stats[:, :n].copy_(ckpt["densify_stats"][:, :n][:, torch.from_numpy(np.ascontiguousarray(order))])

    out = Path(args.model_path)
    if rank == 0:
        out.mkdir(parents=True, exist_ok=True)
        with open(out / "cfg_args.json", "w") as f:
            json.dump({**vars(args), "n_gaussians": n, "n_train_views": len(views), "world_size": world}, f, indent=2)
    save_at, ckpt_at = set(args.save_iterations) | {args.iterations}, set(args.checkpoint_iterations)
    t0 = time.time()
    rollback = Rollback(trainer)
    rollback.take(it0)

    def overflow_anywhere() -> tuple:
        """(some rank's tile lists overflowed, this rank's did)."""
        # ranks render different views and overflow independently, but a rolled-back interval is redone by ALL of them
        over = trainer.rast.overflowed()
        if world > 1:
            import torch.distributed as dist
            if getattr(trainer, "_abi_comm", None) is not None:
                # OMFS_DP_IMPL=abi: the gradient collectives run on the library's own communicator on the compute stream, this
                # flag on torch's.  Two communicators must never have kernels of both enqueued in different orders on different
                # ranks: drain the device first (the call above has read the status word, i.e. synced already -- made explicit)
                torch.cuda.synchronize()
            flag = torch.tensor([1.0 if over else 0.0], device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            return bool(flag.item() > 0), over
        return over, over

    def redo_from_snapshot(it_now: int) -> int:
        anywhere, mine_over = overflow_anywhere()
        if not anywhere:
            return -1
        new_cap = trainer.rast.grow_dup_capacity(2.0) if mine_over else trainer.rast.dup_capacity
        back = rollback.restore()
        if controller is not None:
            controller.log = [e for e in controller.log if e["iteration"] <= back]
        print(f"[ITER {it_now}] rank {rank}: tile-list capacity exceeded after iteration {back} (those iterations rendered empty "
              f"lists): capacity {new_cap} pairs, iterations {back + 1}..{it_now} are redone", flush=True)
        return back

    it = it0
    if trainer.graph_iters != 1:
        raise ValueError("engine/train.py counts ONE iteration per Trainer.step(): graph_iters must stay 1 (a measurement switch of bench.py)")
    while it < args.iterations:
        it += 1
        trainer.step()
        if controller is not None:
            n_before = trainer.model.n
            controller.after_step(it)
            if trainer.model.n != n_before:
                back = redo_from_snapshot(it)          # never snapshot a cloud that was densified on empty renders
                if back >= 0:
                    it = back
                    continue
                rollback.take(it)
                if rank == 0:
                    print(f"[ITER {it}] densify: {controller.log[-1]}", flush=True)
        if it % args.log_every == 0 or it == args.iterations:
            if rank == 0:
                print(f"Training progress: iteration {it}/{args.iterations} loss={trainer.loss_value():.5f} "
                      f"({(it - it0) / (time.time() - t0):.1f} it/s)", flush=True)
            # Tile-list capacity: looked at once per log interval (loss_value() above has synced the host already)
            back = redo_from_snapshot(it)
            if back >= 0:
                it = back
                continue
            rollback.take(it)
        if (it in save_at or it in ckpt_at) and not (it % args.log_every == 0 or it == args.iterations):
            # a save inside a log interval: the interval's overflow check has not run yet, and a file written now would hold
            # parameters stepped on empty renders if the interval is rolled back later (and stay on disk if the run stops
            # before the redo reaches this iteration again) -- so the check (and the redo) come first
            back = redo_from_snapshot(it)
            if back >= 0:
                it = back
                continue
            rollback.take(it)
        if rank == 0 and it in save_at:
            print(f"\n[ITER {it}] Saving Gaussians", flush=True)
            g = trainer.model.to_dict()
            IO.save_gaussian_ply(out / "point_cloud" / f"iteration_{it}" / "point_cloud.ply", g)
            pc_dir = out / "point_cloud" / f"iteration_{it}"
            if trainer.flame_ft is not None:
                np.savez(pc_dir / "flame_param.npz", **trainer.flame_ft.to_flame_params(split["flame"]))
                np.savez(pc_dir / "flame_param_source.npz", **split["flame"])
            else:
                np.savez(pc_dir / "flame_param.npz", **split["flame"])
        if it in ckpt_at:
            trainer.sync_optimizer_state()     # every rank (a collective in the "sharded" exchange): rank 0 saves WHOLE moments
        if rank == 0 and it in ckpt_at:
            print(f"\n[ITER {it}] Saving Checkpoint", flush=True)
            # per-Gaussian columns in the order the cloud was given in (not the trainer's storage order): a resumed run lays
            # the cloud out again and must end up with the same files
            mdl = trainer.model
            torch.save({"iteration": it, "params": mdl.caller_order(mdl.params), "binding": mdl.caller_order(mdl.binding),
                        "adam_m": mdl.caller_order(trainer.opt.m), "adam_v": mdl.caller_order(trainer.opt.v), "sh_degree": trainer.sh_degree,
                        **({"densify_stats": mdl.caller_order(trainer.densify_stats)} if trainer.densify_stats is not None else {}),
                        **({"flame": trainer.flame_ft.state_dict()} if trainer.flame_ft is not None else {})},
                       out / f"chkpnt{it}.pth")
    torch.cuda.synchronize()
    trainer.rast.check_status()
    if rank == 0:
        print(f"\nTraining complete. [{time.time() - t0:.1f} s]")
    trainer.close()
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

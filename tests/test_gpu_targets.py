"""GPU: training targets / gt images (engine/targets.py, omfs_prepare_target) against PIL -- resize by Image.BOX, matte
composited on the run's background -- and, through the engine CLIs, that `gt/` holds what the trainer was shown: with
`fg_masks/` present and `--resolution 2` a render compares with its gt like with like
(`02_Visual_Engine/validation_reporting.py:60-78`, `train_ghost.py:224-240`)."""
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

from omfs_4d_video_gen_amd.engine import synthetic

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


@pytest.mark.parametrize("sw,sh,w,h", [(192, 144, 96, 72), (200, 150, 67, 50), (96, 72, 96, 72), (1920, 1080, 480, 270), (101, 77, 64, 49)])
def test_prepare_target_matches_pil_box_resize_and_matte(sw, sh, w, h):
    from PIL import Image
    from omfs_4d_video_gen_amd.engine import targets as TG
    rng = np.random.default_rng(sw + 7 * w)
    rgb = rng.integers(0, 256, (sh, sw, 3), dtype=np.uint8)
    yy, xx = np.mgrid[:sh, :sw]
    mask = np.clip(255 - 3 * np.hypot(yy - sh / 2, xx - sw / 2) * 255 / max(sh, sw) + 60, 0, 255).astype(np.uint8)   # soft disc
    bg = (1.0, 0.5, 0.25)
    ref_rgb = np.asarray(Image.fromarray(rgb).resize((w, h), Image.BOX)).astype(np.float32)
    ref_m = np.asarray(Image.fromarray(mask).resize((w, h), Image.BOX)).astype(np.float32) / 255.0
    ref = ref_rgb / 255.0 * ref_m[:, :, None] + (1.0 - ref_m[:, :, None]) * np.asarray(bg, np.float32)
    got = TG.prepare_target(rgb, mask, w, h, bg).cpu().numpy().transpose(1, 2, 0)
    # PIL rounds to 8 bits after each of its two passes, the kernel once: one level apart at most (image and matte)
    assert np.abs(got - ref).max() <= 2.0 / 255.0 + 1e-6
    assert np.abs(got - ref).mean() <= 0.35 / 255.0
    if (sw, sh) == (w, h):
        assert np.abs(got - ref).max() <= 1e-6          # no resize: the identity
    got8 = TG.prepare_target(rgb, mask, w, h, bg, as_u8=True).cpu().numpy()
    assert got8.shape == (h, w, 3) and np.abs(got8.astype(np.float32) / 255.0 - got).max() <= 0.5 / 255.0 + 1e-6
    # the matte as the image's alpha channel, and no matte at all
    rgba = np.concatenate([rgb, mask[:, :, None]], 2)
    assert torch.equal(TG.prepare_target(rgba, None, w, h, bg), TG.prepare_target(rgb, mask, w, h, bg))
    plain = TG.prepare_target(rgb, None, w, h, bg).cpu().numpy().transpose(1, 2, 0)
    assert np.abs(plain - ref_rgb / 255.0).max() <= 1.0 / 255.0 + 1e-6


def _psnr(a, b):
    return 10.0 * np.log10(1.0 / max(float(np.mean((a.astype(np.float64) / 255.0 - b.astype(np.float64) / 255.0) ** 2)), 1e-12))


def test_gt_is_the_matted_resized_target_the_trainer_saw(tmp_path, monkeypatch):
    """A dataset whose images lie on BLACK with real mattes in fg_masks/: train_ghost adds --white_background
    (`train_ghost.py:224-240`), the run trains at --resolution 2; render_surgery at 0 mm must then write renders AND gt at the
    training resolution, gt matted on white -- a render is close to its gt and far from the raw (black-background) frame."""
    from omfs_4d_video_gen_amd import render_surgery as rs
    from omfs_4d_video_gen_amd import train_ghost as tg
    from omfs_4d_video_gen_amd.engine import io_formats as IO
    from omfs_4d_video_gen_amd.engine import targets as TG
    from omfs_4d_video_gen_amd.engine.flame_rig import FlameRig
    from omfs_4d_video_gen_amd.engine.trainer import Renderer, View
    monkeypatch.setenv("OMFS_SYNTHETIC_RIG", "1")
    d = tmp_path / "data"
    T, W, H = 60, 192, 144
    rig = synthetic.make_rig(0)
    seq = synthetic.make_flame_sequence(T, 2)
    cams = [synthetic.make_camera(W, H, yaw=0.3 * np.sin(i / 9.0)) for i in range(T)]
    gt_cloud = synthetic.make_gaussians(20000, rig.faces.shape[0], 5)
    r = Renderer(FlameRig.from_synthetic(rig), seq, gt_cloud, W, H, bg=(0.0, 0.0, 0.0))
    imgs, masks = [], []
    for i in range(T):
        imgs.append(r.render(View(cams[i], i), rgb8=True).cpu().numpy().copy())
        masks.append(np.rint((1.0 - r.rast.final_T.cpu().numpy()) * 255.0).astype(np.uint8))
    IO.write_dataset(d, cams, list(range(T)), imgs, seq, fg_masks=True)
    names = sorted(os.listdir(d / "fg_masks"))
    assert len(names) == T
    for name, m in zip(names, masks):                   # real mattes instead of the writer's all-foreground ones
        IO.write_png(d / "fg_masks" / name, m)
    model = tmp_path / "model"
    real_run = subprocess.run
    monkeypatch.setattr(tg.subprocess, "run", lambda cmd, **kw: real_run(list(cmd) + (["--n_gaussians", "20000", "--log_every", "50"]
                                                                                      if str(cmd[1]).endswith("train.py") else []), **kw))
    tg.train(str(d), str(model), iterations=400, resolution=2)
    mod = rs.create_modified_dataset(str(d), 0.0, 0.0)
    renders = Path(rs.render_with_gaussians(str(model), mod))
    gt_dir = renders.parent / "gt"
    split = IO.load_split(str(d), "train")
    ps_gt, ps_raw = [], []
    for k in (0, 7, 23, 41):
        ren, gt = IO.read_png(renders / f"{k:05d}.png"), IO.read_png(gt_dir / f"{k:05d}.png")
        assert ren.shape == (H // 2, W // 2, 3) and gt.shape == ren.shape            # cfg_args.json's resolution is honoured
        rgb, mask = TG.load_frame_pixels(str(d), split["frames"][k])
        want = TG.prepare_target(rgb, mask, W // 2, H // 2, (1.0, 1.0, 1.0), as_u8=True).cpu().numpy()
        assert np.array_equal(gt, want)                                              # gt IS the training target
        assert gt[0, 0].min() >= 250                                                 # matted on white although the frame is on black
        raw_small = TG.prepare_target(rgb, None, W // 2, H // 2, (0.0, 0.0, 0.0), as_u8=True).cpu().numpy()
        ps_gt.append(_psnr(ren, gt)); ps_raw.append(_psnr(ren, raw_small))
    assert min(ps_gt) > 17.0 and np.mean(ps_gt) > np.mean(ps_raw) + 6.0, (ps_gt, ps_raw)

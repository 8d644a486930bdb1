"""Tile pairs D and list segments of the first iterations of a fresh bench trainer (what the pair capacity must hold).
usage (GPU box): python tools/early_d.py [--n_gaussians 300000 --width 1920 --height 1080 --steps 40]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omfs_4d_video_gen_amd.engine import synthetic  # noqa: E402
from omfs_4d_video_gen_amd.engine.flame_rig import FlameRig  # noqa: E402
from omfs_4d_video_gen_amd.engine.trainer import Renderer, Trainer, View  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n_gaussians", type=int, default=300000)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--views", type=int, default=16)
ap.add_argument("--steps", type=int, default=40)
a = ap.parse_args()
N, W, H = a.n_gaussians, a.width, a.height
rig = FlameRig.from_synthetic(synthetic.make_rig(0))
seq = synthetic.make_flame_sequence(max(a.views, 2), 0)
cams = synthetic.make_camera_arc(W, H, a.views)
tr = Renderer(rig, seq, synthetic.make_gaussians(N, rig.n_faces, 1), W, H)
views = []
for i, c in enumerate(cams):
    v = View(c, i % max(a.views, 2))
    v.target = tr.render(v).clone()
    views.append(v)
d_target = int(tr.rast.tile_start[-1])
del tr
t = Trainer(rig, seq, synthetic.make_gaussians(N, rig.n_faces, 0), views, W, H, start_sh_degree=3, finetune_flame=True)
ds, segs = [], []
for _ in range(a.steps):
    t.step()
    torch.cuda.synchronize()
    ds.append(int(t.rast.tile_start[-1]))
    segs.append(int(t.rast.order_seg0[-1]))
t.rast.check_status()
print(json.dumps({"N": N, "size": [W, H], "D_first_steps": ds, "D_max": max(ds), "D_max_per_gaussian": round(max(ds) / N, 2),
                  "segments_max": max(segs), "D_of_the_last_target_render": d_target, "dup_capacity": t.rast.dup_capacity,
                  "seg_capacity": t.rast.seg_capacity}))

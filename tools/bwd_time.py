"""Time the composite backward pass alone on the bench scene after `--pretrain` training steps: the product's omfs_composite_bwd
("dpp") and the second implementations of libomfs_experiments.so ("mfma", "entries"); HIP events around `--reps` launches on a fixed state.
usage (GPU box): python tools/bwd_time.py [--pretrain 200] [--reps 50] [--tag name]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omfs_4d_video_gen_amd import _lib as L  # noqa: E402
from omfs_4d_video_gen_amd.engine import synthetic  # noqa: E402
from omfs_4d_video_gen_amd.engine.flame_rig import FlameRig  # noqa: E402
from omfs_4d_video_gen_amd.engine.trainer import Renderer, Trainer, View  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--pretrain", type=int, default=200)
ap.add_argument("--reps", type=int, default=50)
ap.add_argument("--tag", default="")
ap.add_argument("--impls", default="dpp,mfma,entries")
a = ap.parse_args()
N, W, H = 300000, 1920, 1080
srig = synthetic.make_rig(0)
rig = FlameRig.from_synthetic(srig)
seq = synthetic.make_flame_sequence(16, 0)
cams = synthetic.make_camera_arc(W, H, 16)
tr = Renderer(rig, seq, synthetic.make_gaussians(N, rig.n_faces, 1), W, H)
views = []
for i, c in enumerate(cams):
    v = View(c, i)
    v.target = tr.render(v).clone()
    views.append(v)
del tr
t = Trainer(rig, seq, synthetic.make_gaussians(N, rig.n_faces, 0), views, W, H, start_sh_degree=3, finetune_flame=True)
for _ in range(a.pretrain):
    t.step()
torch.cuda.synchronize()
r, lib, s = t.rast, L.load(), L.stream_ptr()
cam = t._cam(t.view_for_step(t.step_idx - 1), t.sh_degree)
gb = L.GradBuffersC(L.ptr(r.dsplat), L.ptr(t.grads), L.ptr(r.dimage), 0, 0, 0)
out = {"tag": a.tag, "D": int(r.tile_start[-1])}
ref = None
for impl in a.impls.split(","):
    times = []
    for k in range(a.reps + 5):
        r.dsplat.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.composite_bwd(impl, cam, r.rb, gb, s)
        e1.record()
        torch.cuda.synchronize()
        if k >= 5:
            times.append(e0.elapsed_time(e1))
    ds = r.dsplat.clone()
    if ref is None:
        ref = ds
    else:
        out["max_rel_diff_vs_" + a.impls.split(",")[0]] = float((ds - ref).abs().max() / ref.abs().max())
    times.sort()
    out[impl] = {"ms_mean": round(sum(times) / len(times), 4), "ms_min": round(times[0], 4), "ms_median": round(times[len(times) // 2], 4)}
print(json.dumps(out))

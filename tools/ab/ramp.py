"""Is the falling per-step time of the first 25 iterations the GPU warming up or the scene changing?  Trainer A runs 300 steps, then a
fresh trainer B (same initial state) is timed per step."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from omfs_4d_video_gen_amd.engine import synthetic
from omfs_4d_video_gen_amd.engine.flame_rig import FlameRig
from omfs_4d_video_gen_amd.engine.trainer import Renderer, Trainer, View
N, W, H = 300000, 1920, 1080
rig = FlameRig.from_synthetic(synthetic.make_rig(0))
seq = synthetic.make_flame_sequence(16, 0)
cams = synthetic.make_camera_arc(W, H, 16)
tr = Renderer(rig, seq, synthetic.make_gaussians(N, rig.n_faces, 1), W, H)
views = []
for i, c in enumerate(cams):
    v = View(c, i); v.target = tr.render(v).clone(); views.append(v)
del tr
def run(n, tag):
    t = Trainer(rig, seq, synthetic.make_gaussians(N, rig.n_faces, 0), views, W, H, start_sh_degree=3, finetune_flame=True)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        t.step(); ev[i + 1].record()
    torch.cuda.synchronize()
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
    print(tag, " ".join(f"{x:.3f}" for x in ms[:26]), "| last 10 mean", round(sum(ms[-10:]) / 10, 4), "| pairs", int(t.rast.tile_pairs()) if hasattr(t.rast, "tile_pairs") else "")
run(300, "A")
run(40, "B")

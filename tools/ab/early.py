"""Stage times of iterations 5..24 of a fresh run (what `bench.py --steps 20 --warmup 5` times) against iterations 300..339."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from omfs_4d_video_gen_amd.engine import synthetic
from omfs_4d_video_gen_amd.engine.flame_rig import FlameRig
from omfs_4d_video_gen_amd.engine.trainer import Renderer, Trainer, View, StageTimer
N, W, H = 300000, 1920, 1080
rig = FlameRig.from_synthetic(synthetic.make_rig(0))
seq = synthetic.make_flame_sequence(16, 0)
cams = synthetic.make_camera_arc(W, H, 16)
tr = Renderer(rig, seq, synthetic.make_gaussians(N, rig.n_faces, 1), W, H)
views = []
for i, c in enumerate(cams):
    v = View(c, i); v.target = tr.render(v).clone(); views.append(v)
del tr
t = Trainer(rig, seq, synthetic.make_gaussians(N, rig.n_faces, 0), views, W, H, start_sh_degree=3, finetune_flame=True)
def timed(n):
    t.timer = StageTimer(True)
    for _ in range(n): t.step()
    torch.cuda.synchronize()
    s = t.timer.summary(); t.timer = StageTimer(False)
    return {k: round(v[0], 4) for k, v in s.items()}
for _ in range(5): t.step()
a = timed(20)
for _ in range(275): t.step()
b = timed(40)
print("stage            it 5-24   it 300-339")
for k in a: print(f"{k:16s} {a[k]:8.4f} {b[k]:8.4f}  {a[k]-b[k]:+.4f}")
print("sum", round(sum(a.values()), 4), round(sum(b.values()), 4))

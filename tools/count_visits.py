"""Debug build only (-DOMFS_DEBUG_COUNTERS): visits / visits with a hit / hit lanes of the composite kernels."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omfs_4d_video_gen_amd import _lib as L
from omfs_4d_video_gen_amd.engine import synthetic
from omfs_4d_video_gen_amd.engine.flame_rig import FlameRig
from omfs_4d_video_gen_amd.engine.trainer import Renderer, Trainer, View
W, H, N = 1920, 1080, 300000
srig = synthetic.make_rig(0); rig = FlameRig.from_synthetic(srig); seq = synthetic.make_flame_sequence(16, 0)
cams = synthetic.make_camera_arc(W, H, 16); g0 = synthetic.make_gaussians(N, rig.n_faces, 0); g1 = synthetic.make_gaussians(N, rig.n_faces, 1)
tr = Renderer(rig, seq, g1, W, H); views = []
for i, c in enumerate(cams):
    v = View(c, i); v.target = tr.render(v).clone(); views.append(v)
t = Trainer(rig, seq, g0, views, W, H, start_sh_degree=3)
for _ in range(20): t.step()
torch.cuda.synchronize()
lib = L.load()._lib if hasattr(L.load(), "_lib") else L.load()
buf = (ctypes.c_ulonglong * 32)()
cd = ctypes.CDLL(L.LIB_PATH)
cd.omfs_debug_counters(buf, 1)
for _ in range(16): t.step()
torch.cuda.synchronize()
cd.omfs_debug_counters(buf, 0)
b = [x / 16 for x in buf]
print("bwd: visits %.0f, with hit %.0f (%.1f%%), hit lanes per hit-visit %.1f" % (b[0], b[1], 100 * b[1] / b[0], b[2] / max(b[1], 1)))
print("fwd(F1): visits %.0f, with hit %.0f (%.1f%%), hit lanes per hit-visit %.1f" % (b[3], b[4], 100 * b[4] / b[3], b[5] / max(b[4], 1)))
print("bwd per-step: union %.0f, max_row %.0f (%.3f of union), sum_rows %.0f (k = %.2f sub-blocks per visit); per-wave max row total %.0f (%.3f of union); waves %.0f"
      % (b[8], b[9], b[9] / b[8], b[10], b[10] / b[8], b[11], b[11] / b[8], b[14]))
print("bwd: geometric hit lanes per visit %.1f, hit lanes per visit %.1f, sub-blocks with a hit per visit %.2f" % (b[12] / b[0], b[2] / b[0], b[13] / b[0]))
print("fwd: real visits %.0f, sub-blocks with a hit per visit %.2f, flagged live sub-blocks per visit %.2f, hit lanes %.1f"
      % (b[16], b[17] / b[16], b[18] / b[16], b[5] / b[16]))
print("bwd visits by number of sub-blocks still active (1..4): " + ", ".join("%d: %.1f%% (%.1f hit lanes)" % (k, 100 * b[19 + k] / b[0], b[23 + k] / max(b[19 + k], 1)) for k in (1, 2, 3, 4)))
print("fwd visits by number of live sub-blocks (1..4): " + ", ".join("%d: %.1f%%" % (k, 100 * b[27 + k] / b[16]) for k in (1, 2, 3, 4)))

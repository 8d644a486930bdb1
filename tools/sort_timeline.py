"""Debug build only (EXTRA_HIPCC_FLAGS=-DOMFS_DEBUG_TIMELINE): per-workgroup timeline of the tile-sort launch (two launches up to
round 2) of one training step of the bench scene -- span, duration against list length, which path each list took
(1 = bucket sort with the copy in LDS, 2 = bucket sort through keys_tmp, 3 = radix fallback).
usage (GPU box): EXTRA_HIPCC_FLAGS=-DOMFS_DEBUG_TIMELINE bash omfs_4d_video_gen_amd/csrc/build.sh && python tools/sort_timeline.py"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omfs_4d_video_gen_amd import _lib as L
from omfs_4d_video_gen_amd.engine import synthetic
from omfs_4d_video_gen_amd.engine.flame_rig import FlameRig
from omfs_4d_video_gen_amd.engine.trainer import Renderer, Trainer, View

W, H, N = 1920, 1080, 300000
srig = synthetic.make_rig(0); rig = FlameRig.from_synthetic(srig); seq = synthetic.make_flame_sequence(16, 0)
cams = synthetic.make_camera_arc(W, H, 16)
tr = Renderer(rig, seq, synthetic.make_gaussians(N, rig.n_faces, 1), W, H); views = []
for i, c in enumerate(cams):
    v = View(c, i); v.target = tr.render(v).clone(); views.append(v)
t = Trainer(rig, seq, synthetic.make_gaussians(N, rig.n_faces, 0), views, W, H, start_sh_degree=3)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    t.step()
torch.cuda.synchronize()
cd = ctypes.CDLL(L.LIB_PATH)
buf = (ctypes.c_ulonglong * (3 * 16384))()
cd.omfs_debug_sort_timeline(0, buf, 1)
t.step()
torch.cuda.synchronize()
for cls, name in ((0, "1024-thread workgroups"), (1, "512-thread workgroups (rounds 1-2 only)")):
    assert cd.omfs_debug_sort_timeline(cls, buf, 0) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(3, 16384).astype(np.int64)
    ok = a[0] > 0
    if not ok.any():
        print(name, ": nothing recorded"); continue
    t0, t1, n, path = a[0][ok], a[1][ok], a[2][ok] & 0xFFFFFFFF, a[2][ok] >> 32
    dur = (t1 - t0) * 10e-3
    span = (t1.max() - t0.min()) * 10e-3
    fit = np.polyfit(n, dur, 1)
    print(f"{name}: {ok.sum()} lists, span {span:.1f} us, entries {n.sum()}, duration us p50 {np.percentile(dur, 50):.2f} p99 {np.percentile(dur, 99):.2f} "
          f"max {dur.max():.2f} (list of {n[np.argmax(dur)]}, path {path[np.argmax(dur)]}); fit {fit[1]:.2f} us + {fit[0] * 1e3:.2f} ns per entry; "
          f"paths {dict(zip(*np.unique(path, return_counts=True)))}")
    edges = np.linspace(t0.min(), t1.max(), 11)
    print("   workgroups in flight per tenth of the span:", [int(((t0 < edges[i + 1]) & (t1 > edges[i])).sum()) for i in range(10)])
    for lo, hi in ((2, 1024), (1024, 4096), (4096, 7936), (7936, 1 << 30)):
        sel = (n > lo) & (n <= hi)
        if sel.any():
            st, en = (t0[sel] - t0.min()) * 10e-3, (t1[sel] - t0.min()) * 10e-3
            print(f"   lists of {lo + 1}..{hi if hi < 1 << 30 else 'inf'} pairs: {int(sel.sum())}, duration us p50 {np.percentile(dur[sel], 50):.1f} max {dur[sel].max():.1f}; "
                  f"start p50 {np.percentile(st, 50):.1f} max {st.max():.1f}; end p50 {np.percentile(en, 50):.1f} p99 {np.percentile(en, 99):.1f} max {en.max():.1f}")

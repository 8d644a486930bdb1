"""Debug build only (EXTRA_HIPCC_FLAGS=-DOMFS_DEBUG_TIMELINE): per-wave start / end / visits of the three implementations
of the composite backward (the product kernel; matrix-core reduction and lanes = entries from libomfs_experiments.so) on the bench scene: how many waves do real work, how long they live, what a visit costs.
usage (GPU box): python tools/bwd_timeline.py [--pretrain 100]"""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omfs_4d_video_gen_amd import _lib as L  # noqa: E402
from omfs_4d_video_gen_amd.engine import synthetic  # noqa: E402
from omfs_4d_video_gen_amd.engine.flame_rig import FlameRig  # noqa: E402
from omfs_4d_video_gen_amd.engine.trainer import Renderer, Trainer, View  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--pretrain", type=int, default=100)
a = ap.parse_args()
W, H, N = 1920, 1080, 300000
srig = synthetic.make_rig(0); rig = FlameRig.from_synthetic(srig); seq = synthetic.make_flame_sequence(16, 0)
cams = synthetic.make_camera_arc(W, H, 16)
tr = Renderer(rig, seq, synthetic.make_gaussians(N, rig.n_faces, 1), W, H); views = []
for i, c in enumerate(cams):
    v = View(c, i); v.target = tr.render(v).clone(); views.append(v)
del tr
t = Trainer(rig, seq, synthetic.make_gaussians(N, rig.n_faces, 0), views, W, H, start_sh_degree=3, finetune_flame=True)
for _ in range(a.pretrain):
    t.step()
torch.cuda.synchronize()
r, lib, s = t.rast, L.load(), L.stream_ptr()
cam = t._cam(t.view_for_step(t.step_idx - 1), t.sh_degree)
gb = L.GradBuffersC(L.ptr(r.dsplat), L.ptr(t.grads), L.ptr(r.dimage), 0, 0, 0)
cd = ctypes.CDLL(L.LIB_PATH)
NTL = 1 << 19
buf = (ctypes.c_ulonglong * (3 * NTL))()
L.load_experiments()
ce = ctypes.CDLL(L.EXPERIMENTS_PATH)               # the second implementations stamp the arrays of THEIR library
for impl in ("dpp", "mfma", "entries"):
    read = cd.omfs_debug_timeline if impl == "dpp" else ce.omfs_experiment_debug_timeline
    for _ in range(3):
        L.composite_bwd(impl, cam, r.rb, gb, s)
    torch.cuda.synchronize()
    read(2, buf, NTL, 1)          # reset
    L.composite_bwd(impl, cam, r.rb, gb, s)
    torch.cuda.synchronize()
    assert read(2, buf, NTL, 2) == 0
    raw = np.frombuffer(buf, dtype=np.uint64)
    tt = raw[:2 * NTL].reshape(2, NTL).astype(np.int64)
    work = raw[2 * NTL:2 * NTL + NTL // 2].view(np.uint32)[:NTL].astype(np.int64)
    ok = tt[0] > 0
    t0, t1, w = tt[0][ok], tt[1][ok], work[ok]
    dur = (t1 - t0) * 10e-3                          # us (100 MHz counter)
    span = (t1.max() - t0.min()) * 10e-3
    real = w > 0
    mid = (~real) & (dur > 0.5)
    print(f"[{impl}] span {span:.1f} us; {t0.size} waves recorded, {int(real.sum())} visited >= 1 splat ({int(w.sum())} visits, "
          f"{w[real].mean():.1f} per working wave, p50 {np.percentile(w[real], 50):.0f} p90 {np.percentile(w[real], 90):.0f} max {w.max()})")
    print(f"   working waves: lifetime us p10 {np.percentile(dur[real], 10):.2f} p50 {np.percentile(dur[real], 50):.2f} p90 {np.percentile(dur[real], 90):.2f} "
          f"max {dur[real].max():.2f}; sum {dur[real].sum() / (span * 1024):.2f} resident per SIMD over the span")
    print(f"   waves without a visit: {int((~real).sum())}, lifetime p50 {np.percentile(dur[~real], 50):.2f} p90 {np.percentile(dur[~real], 90):.2f} us, "
          f"sum {dur[~real].sum() / (span * 1024):.2f} resident per SIMD")
    fit = np.polyfit(w[real], dur[real], 1)
    print(f"   lifetime of a working wave = {fit[0] * 1e3:.0f} ns per visit + {fit[1]:.2f} us")
    for lo, hi in ((1, 4), (4, 16), (16, 48), (48, 128), (128, 100000)):
        sel = real & (w >= lo) & (w < hi)
        if sel.any():
            print(f"   {lo:4d} <= visits < {hi:6d}: {int(sel.sum()):6d} waves, {int(w[sel].sum()):8d} visits, lifetime mean {dur[sel].mean():6.2f} us, "
                  f"{1e3 * dur[sel].sum() / w[sel].sum():6.0f} ns per visit")
    edges = np.linspace(t0.min(), t1.max(), 11)
    print("   working waves in flight per tenth of the span:", [int(((t0[real] < edges[i + 1]) & (t1[real] > edges[i])).sum()) for i in range(10)])
    print("   working waves STARTED per tenth of the span:  ", [int(((t0[real] >= edges[i]) & (t0[real] < edges[i + 1])).sum()) for i in range(10)])
    # who makes the tail: visits and lifetimes by start time, and the last finishers
    rows = []
    for i in range(10):
        sel = real & (t0 >= edges[i]) & (t0 < edges[i + 1])
        rows.append(f"{int(w[sel].mean()) if sel.any() else 0}/{dur[sel].mean() if sel.any() else 0:.0f}")
    print("   mean visits / mean lifetime (us) of the working waves by the tenth they START in:", rows)
    order = np.argsort(t1)[::-1]
    lastn = order[:500]
    st = (t0[lastn] - t0.min()) * 10e-3
    print(f"   the 500 last finishers: end {((t1[lastn] - t0.min()) * 10e-3).min():.0f}-{span:.0f} us; started at p10 {np.percentile(st, 10):.0f} p50 {np.percentile(st, 50):.0f} "
          f"p90 {np.percentile(st, 90):.0f} us; visits p10 {np.percentile(w[lastn], 10):.0f} p50 {np.percentile(w[lastn], 50):.0f} p90 {np.percentile(w[lastn], 90):.0f}; "
          f"lifetime p50 {np.percentile(dur[lastn], 50):.0f} us")
    idx = np.nonzero(ok)[0]
    print(f"   their position in the grid (wave index / 1000): p10 {np.percentile(idx[lastn], 10) / 1e3:.0f} p50 {np.percentile(idx[lastn], 50) / 1e3:.0f} "
          f"p90 {np.percentile(idx[lastn], 90) / 1e3:.0f} of {idx.max() / 1e3:.0f}; working waves lie at p50 {np.percentile(idx[real], 50) / 1e3:.0f} p99 {np.percentile(idx[real], 99) / 1e3:.0f}")

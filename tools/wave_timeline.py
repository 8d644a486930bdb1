"""Debug build only (EXTRA_HIPCC_FLAGS=-DOMFS_DEBUG_TIMELINE): when do the waves of the three composite kernels run?
Per kernel: span of the launch, number of waves that did work, sum of wave durations / (span x 1024 SIMDs) = resident
working waves per SIMD, duration percentiles, and the number of waves in flight in ten slices of the span.
usage (GPU box): EXTRA_HIPCC_FLAGS=-DOMFS_DEBUG_TIMELINE bash omfs_4d_video_gen_amd/csrc/build.sh && python tools/wave_timeline.py"""
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
for _ in range(20):
    t.step()
torch.cuda.synchronize()
cd = ctypes.CDLL(L.LIB_PATH)
NTL = 1 << 19
buf = (ctypes.c_ulonglong * (3 * NTL))()
cd.omfs_debug_timeline(0, buf, NTL, 1)
_clr = (ctypes.c_uint32 * (16 * NTL))()
cd.omfs_debug_timeline(0, ctypes.cast(_clr, ctypes.POINTER(ctypes.c_ulonglong)), NTL, 4)      # clears the per-step table
t.step()
torch.cuda.synchronize()
for k, name, waves_per in ((0, "composite_fwd (one wave per entry)", 1), (1, "composite_fwd_deep (8-wave workgroups)", 8), (2, "composite_bwd", 1)):
    assert cd.omfs_debug_timeline(k, buf, NTL, 2) == 0       # time rows + work counters
    raw = np.frombuffer(buf, dtype=np.uint64)
    a = raw[:2 * NTL].reshape(2, NTL).astype(np.int64)
    work = raw[2 * NTL:2 * NTL + NTL // 2].view(np.uint32)[:NTL].astype(np.int64)
    ok = a[0] > 0
    t0, t1 = a[0][ok], a[1][ok]
    if t0.size == 0:
        print(name, ": nothing recorded"); continue
    span = (t1.max() - t0.min()) * 10e-3          # us (100 MHz counter)
    dur = (t1 - t0) * 10e-3
    busy = dur.sum() * waves_per / (span * 1024)
    edges = np.linspace(t0.min(), t1.max(), 11)
    inflight = [int(((t0 < edges[i + 1]) & (t1 > edges[i])).sum()) * waves_per for i in range(10)]
    print(f"{name}: span {span:.1f} us, {t0.size} recorded, resident working waves/SIMD {busy:.2f}, "
          f"duration us p50 {np.percentile(dur, 50):.2f} p90 {np.percentile(dur, 90):.2f} p99 {np.percentile(dur, 99):.2f} max {dur.max():.2f}")
    print("   waves in flight per tenth of the span:", inflight)
    w = work[ok]
    if w.max() > 0:
        big = dur > np.percentile(dur, 99)
        fit = np.polyfit(w[big], dur[big], 1)
        print(f"   longest 1 %: {int(big.sum())} waves, splats visited p50 {np.percentile(w[big], 50):.0f} max {w[big].max()}, "
              f"duration = {fit[0] * 1e3:.0f} ns per visited splat + {fit[1]:.1f} us")
    if k == 0:
        # phase split of the one-wave forward (shader-clock cycles): waiting for the gather, staging, walking, the rest
        pbuf = (ctypes.c_uint32 * (4 * NTL))()
        assert cd.omfs_debug_timeline(0, ctypes.cast(pbuf, ctypes.POINTER(ctypes.c_ulonglong)), NTL, 3) == 0
        ph = np.frombuffer(pbuf, dtype=np.uint32).reshape(4, NTL).astype(np.float64)[:, ok]
        tot = ph.sum(0)
        mhz = np.median(tot[dur > 20] / dur[dur > 20]) if (dur > 20).any() else float("nan")
        print(f"   shader clock against the real-time counter: {mhz:.0f} cycles per us")
        for name_, sel in (("all waves", np.ones(big.shape, bool)), ("longest 1 %", big), ("start in the first tenth", t0 < edges[1])):
            q = ph[:, sel].sum(1)
            print(f"   {name_:24s}: wait for gather {q[0] / q.sum():.2f}  stage {q[1] / q.sum():.2f}  walk {q[2] / q.sum():.2f}  other {q[3] / q.sum():.2f}"
                  f"   (mean {ph[:, sel].sum(0).mean() / mhz:.1f} us per wave)")
        sbuf = (ctypes.c_uint32 * (16 * NTL))()
        assert cd.omfs_debug_timeline(0, ctypes.cast(sbuf, ctypes.POINTER(ctypes.c_ulonglong)), NTL, 4) == 0
        st = np.frombuffer(sbuf, dtype=np.uint32).reshape(2, 8, NTL).astype(np.float64)[:, :, ok][:, :, big]
        full = st[0, 7] > 0                       # long waves that walked all eight 64-entry steps
        if full.any():
            tt, ww = st[0][:, full] * 10e-3, st[1][:, full]
            start = (t0[big][full] - t0.min()) * 10e-3
            print(f"   {int(full.sum())} of the longest waves walked 8 steps; per step: end time since the launch began (us), duration (us), entries walked, ns per entry")
            prev_t, prev_w = np.zeros(tt.shape[1]), np.zeros(tt.shape[1])
            for s_ in range(8):
                d, e = tt[s_] - prev_t, ww[s_] - prev_w
                print(f"      step {s_}: ends {np.mean(start + tt[s_]):6.1f}  lasts {d.mean():5.1f}  entries {e.mean():5.1f}  {1e3 * d.sum() / max(e.sum(), 1):5.0f} ns/entry")
                prev_t, prev_w = tt[s_], ww[s_]
        lw = ph[:, big]
        print(f"   longest 1 %: per visited splat {np.median(lw[2] / np.maximum(w[big], 1)):.0f} walk cycles; per 64-entry step "
              f"{np.median(lw[1] / np.maximum(np.ceil(w[big] / 50.0), 1)):.0f} staging cycles (assuming ~50 visits per step)")

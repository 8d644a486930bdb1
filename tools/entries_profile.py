"""Debug build only (EXTRA_HIPCC_FLAGS="-DOMFS_DEBUG_COUNTERS -DOMFS_DEBUG_TIMELINE"): what the lanes = list-entries backward
(composite_bwd_entries_kernel, libomfs_experiments.so) does on the bench scene -- units, passes, streamed pixels, wave-steps, hit
lanes per step (the density the design lives on), and its per-wave timeline (working waves, lifetimes, residency).
usage (GPU box): EXTRA_HIPCC_FLAGS="-DOMFS_DEBUG_COUNTERS -DOMFS_DEBUG_TIMELINE" bash omfs_4d_video_gen_amd/csrc/build.sh && python tools/entries_profile.py"""
import argparse
import ctypes
import json
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
ap.add_argument("--mode", default="both", choices=("both", "counters", "timeline"), help="what the experiments library was built with: the per-step counters slow the kernel a hundredfold, so the timeline is taken from a build without them")
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
r, s = t.rast, L.stream_ptr()
cam = t._cam(t.view_for_step(t.step_idx - 1), t.sh_degree)
gb = L.GradBuffersC(L.ptr(r.dsplat), L.ptr(t.grads), L.ptr(r.dimage), 0, 0, 0)
L.load_experiments()
ce = ctypes.CDLL(L.EXPERIMENTS_PATH)
cnt = (ctypes.c_ulonglong * 32)()
for _ in range(2):
    L.composite_bwd("entries", cam, r.rb, gb, s)
torch.cuda.synchronize()
NTL = 1 << 19
buf = (ctypes.c_ulonglong * (3 * NTL))()
if a.mode != "timeline":
    ce.omfs_experiment_debug_counters(cnt, 1)
if a.mode != "counters":
    ce.omfs_experiment_debug_timeline(2, buf, NTL, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
L.composite_bwd("entries", cam, r.rb, gb, s)
e1.record()
torch.cuda.synchronize()
if a.mode != "timeline":
    ce.omfs_experiment_debug_counters(cnt, 0)
if a.mode != "counters":
    assert ce.omfs_experiment_debug_timeline(2, buf, NTL, 2) == 0
b = [int(x) for x in cnt]
units, passes, pix, ents, wsteps, hit_steps, hit_lanes = b[14], b[15], b[8], b[9], b[0], b[1], b[2]
# the same scene seen by the product kernel: visited entries per tile (depth), pixels
nc = r.n_contrib
gy, gx = r.gy, r.gx
pad = torch.zeros(gy * 16, gx * 16, dtype=torch.int32, device="cuda")
pad[:H, :W] = nc
depth = pad.view(gy, 16, gx, 16).permute(0, 2, 1, 3).reshape(gy * gx, 256).max(1).values
raw = np.frombuffer(buf, dtype=np.uint64)
tt = raw[:2 * NTL].reshape(2, NTL).astype(np.int64)
work = raw[2 * NTL:2 * NTL + NTL // 2].view(np.uint32)[:NTL].astype(np.int64)
ok = tt[0] > 0
if not ok.any():
    ok[:] = True
t0, t1, w = tt[0][ok], tt[1][ok], work[ok]
dur = (t1 - t0) * 10e-3
span = max((t1.max() - t0.min()) * 10e-3, 1e-9)
real = w > 0
if not real.any():
    real[:] = True
edges = np.linspace(t0.min(), t1.max(), 11)
out = {
    "kernel": "composite_bwd_entries_kernel (debug-counter + timeline build: slower than the product build)",
    "launch_ms_debug_build": round(e0.elapsed_time(e1), 4), "D": int(r.tile_start[-1]), "D_visit_tile_level": int(depth.sum()),
    "units_with_work": units, "passes": passes, "streamed_pixels_per_unit": round(pix / max(units, 1), 1),
    "entries_per_unit": round(ents / max(units, 1), 1), "wave_steps": wsteps, "lane_steps": wsteps * 64,
    "wave_steps_with_a_hit": hit_steps, "hit_lanes": hit_lanes, "hit_lanes_per_wave_step": round(hit_lanes / max(wsteps, 1), 2),
    "fill_share_of_steps": round(1.0 - (pix * passes / max(units, 1)) / max(wsteps, 1), 3) if units else None,
    "timeline": {"span_us": round(float(span), 1), "waves_recorded": int(t0.size), "working_waves": int(real.sum()),
                 "steps_per_working_wave": round(float(w[real].mean()), 1), "lifetime_us_p50": round(float(np.percentile(dur[real], 50)), 2),
                 "lifetime_us_p90": round(float(np.percentile(dur[real], 90)), 2), "lifetime_us_max": round(float(dur[real].max()), 2),
                 "ns_per_wave_step": round(float(1e3 * dur[real].sum() / w[real].sum()), 1),
                 "resident_working_waves_per_simd": round(float(dur[real].sum() / (span * 1024)), 2),
                 "working_waves_in_flight_per_tenth_of_the_span": [int(((t0[real] < edges[i + 1]) & (t1[real] > edges[i])).sum()) for i in range(10)]},
    "mode": a.mode,
}
print(json.dumps(out))

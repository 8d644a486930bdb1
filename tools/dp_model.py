#!/usr/bin/env python3
"""The data-parallel step model of DESIGN.md section 7, as a script: from the measured single-GPU step and the measured fixed cost of
each exchange mode on a one-rank RCCL group (profiles/r05_dp_overhead.json) to the step time, the aggregate it/s and the scaling
at W = 2 / 4 / 8 as a function of the ONE unknown -- the RCCL bus bandwidth the node delivers for a 13.2 MB (compact) or 70.8 MB
(full) all-reduce -- and the thresholds the first SCALE record is to be read against.  Nothing here is measured on more than one GPU.

  python tools/dp_model.py [profiles/r05_dp_overhead.json] [--single_ms 0.803]

Definitions.  busbw B = the figure rccl-tests prints: an all-reduce of S bytes takes S * 2 (W - 1) / W / B (+ latency), an all-gather in
which every rank contributes s bytes takes s * (W - 1) / B (+ latency).  xGMI: 7 links x 76.8 GB/s per direction and GPU; a ring
uses one link per hop (B <= 76.8), the direct all-to-all algorithms use W - 1 of them (B <= 76.8 (W - 1))."""
import argparse
import json
import os

ap = argparse.ArgumentParser()
ap.add_argument("overhead", nargs="?", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r05_dp_overhead.json"))
ap.add_argument("--latency_us", type=float, default=25.0)
ap.add_argument("--n_pad", type=int, default=300032)
a = ap.parse_args()
rec = json.load(open(a.overhead))
m = rec["modes"]
single = m["plain"]["ms_per_step"] * 1e3                     # the one-GPU line (in-place SH Adam), us
fixed = {k: m[k]["ms_per_step"] * 1e3 for k in ("compact", "full", "sharded") if k in m}      # whole step with the exchange's launches, no link time
S14, S59, s3 = 11 * a.n_pad * 4.0, 59 * a.n_pad * 4.0, 3 * a.n_pad * 4.0     # (S14: the low planes of the compact exchange -- 11 since round 5)
# us of work the asynchronous collectives run under (engine/trainer.py, data-parallel step with FLAME fine-tuning on, the bench workload):
# the all-gather of dL/dcolour is issued behind composite_bwd and waited for behind project_bwd (42 us) AND the FLAME backward (33 us: its
# gradients travel in the all-reduce, so it runs first); the 11-plane all-reduce is issued there and waited for behind the rebuilt-plane
# Adam launch (75 us)
HIDE_AG = 42.0 + 33.0
# the rebuild of the 45 SH planes + their Adam pass, timed alone with W views (profiles/r05_fold_time.json, "two_launches"): the work the
# all-reduce runs under -- and, beyond its one-view figure, a cost of the step that the one-rank measurement of `fixed` does not contain
REBUILD_ADAM_US = {1: 68.9, 2: 73.9, 4: 74.7, 8: 91.6}
lat = a.latency_us


def step_us(mode, W, B):            # B in GB/s = 1e3 bytes/us
    f = 2.0 * (W - 1) / W
    if mode == "compact":
        ar = S14 * f / (B * 1e3) + lat
        ag = s3 * (W - 1) / (B * 1e3) + lat
        return fixed["compact"] + (REBUILD_ADAM_US[W] - REBUILD_ADAM_US[1]) + max(0.0, ar - REBUILD_ADAM_US[W]) + max(0.0, ag - HIDE_AG)
    if mode == "full":
        return fixed["full"] + S59 * f / (B * 1e3) + lat
    # sharded: reduce-scatter of the gradients + all-gather of the parameters = the bytes of one all-reduce; Adam on 1/W of the elements
    adam = 72.0
    return fixed["sharded"] - adam * (1.0 - 1.0 / W) + S59 * f / (B * 1e3) + 2 * lat


def bisect(fn, lo=1.0, hi=5000.0):
    for _ in range(60):
        mid = 0.5 * (lo + hi)
        if fn(mid):
            hi = mid
        else:
            lo = mid
    return hi


out = {"single_gpu_step_us": round(single, 1), "fixed_step_us_one_rank_group": {k: round(v, 1) for k, v in fixed.items()},
       "bytes": {"allreduce_11_planes": S14, "allreduce_59_planes": S59, "allgather_per_rank": s3}, "latency_us_per_collective": lat, "per_W": {}}
for W in (2, 4, 8):
    peak = 76.8 * (W - 1)
    row = {"xgmi_peak_busbw_GBs": round(peak, 1)}
    for eff in (0.7, 0.5, 0.35):
        B = eff * peak
        t = {mo: step_us(mo, W, B) for mo in fixed}
        best = min(t, key=t.get)
        row[f"at_{eff:.2f}_of_peak"] = {"busbw_GBs": round(B, 1), **{mo: {"step_us": round(v, 1), "its": round(W * 1e6 / v), "x_single": round(W * single / v, 2)} for mo, v in t.items()},
                                        "fastest": best}
    # thresholds: the busbw below which ... (compact exchange)
    row["busbw_GBs_hiding_the_11_plane_allreduce"] = round(bisect(lambda B: S14 * 2 * (W - 1) / W / (B * 1e3) + lat <= REBUILD_ADAM_US[W]), 1)
    row["busbw_GBs_below_which_full_beats_compact"] = None      # never on xGMI: see note
    target = {2: 1.5, 4: 3.0, 8: 6.0}[W]
    row[f"busbw_GBs_needed_for_{target}x"] = round(bisect(lambda B: W * single / step_us("compact", W, B) >= target), 1)
    out["per_W"][str(W)] = row
out["note"] = ("full never beats compact on xGMI: its fixed cost is %.0f us lower but it moves %.0f MB more per all-reduce, which at the W = 8 peak "
               "of 537.6 GB/s already costs %.0f us; the mode to switch to when the measured busbw is LOW is none of the three -- below the "
               "threshold of the last column the target is lost in every mode" % (fixed["compact"] - fixed["full"], (S59 - S14) / 1e6, (S59 - S14) * 1.75 / 537.6e3))
print(json.dumps(out, indent=1))

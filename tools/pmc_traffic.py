#!/usr/bin/env python3
"""profiles/traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of tools/run_scene.py --train.

The counters are in KiB.  What they count was calibrated on known byte counts (tools/micro/fetch_calib.hip ->
profiles/r03_fetch_calibration.json, MI355X): FETCH_SIZE reports HALF of the bytes of coalesced streaming reads (4, 8 or 16 B
per lane alike -- MI355X_MICROARCH.md's gfx950 correction), but for 16-byte gathers of random records it reports a whole
64-byte sector per record (i.e. at least the bytes asked for, never half), 64-byte records exactly, and WRITE_SIZE counts a
64-byte request per float-atomic record.  So the x2 applies to the STREAMING kernels only; for the gather-bound composite
kernels the raw figure is the traffic.  Every stage is written three ways: `<stage>@<tag>` = the calibrated estimate,
`...:raw` = FETCH + WRITE as counted, `...:fetch_x2` = 2 FETCH + WRITE (what rounds 1-2 reported for every kernel).
The run's tile-pair count D is stored as `_D` (the traffic of the list-walking kernels scales with it).  Kernels are grouped
into the stage names bench.py reports."""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import summarise  # noqa: E402

# stage -> [(kernel name fragment, FETCH_SIZE factor)]: 2 = coalesced streaming reads dominate, 1 = gathers of 16- / 64-byte records
STAGES = {
    "composite_fwd": [("composite_fwd_kernel", 1.0), ("composite_fwd_deep_kernel", 1.0)],
    "composite_bwd": [("composite_bwd_kernel", 1.0)],
    "loss": [("ssim_fwd_kernel", 2.0), ("ssim_bwd_kernel", 2.0), ("loss_reduce_kernel", 2.0)],
    "tile_sort": [("tile_sort_kernel", 2.0)],
    "bin_count": [("bin_count_kernel", 2.0)],
    "bin_scan": [("tile_scan_kernel", 2.0)],
    "bin_scatter": [("bin_scatter_kernel", 2.0)],
    "project": [("project_fwd_kernel", 2.0)],
    "project_bwd": [("project_bwd_kernel", 2.0), ("count_visible_kernel", 2.0)],
    "adam": [("adam_kernel", 2.0), ("adam_sh_rest_kernel", 2.0)],
    "flame": [("flame_joints_kernel", 2.0), ("flame_lbs_kernel", 2.0), ("flame_pose_lbs_kernel", 2.0), ("face_frames_kernel", 1.0)],
    "flame_bwd": [("face_frames_bwd_kernel", 1.0), ("flame_skin_bwd_kernel", 2.0), ("basis_t_gemv_kernel", 2.0),
                  ("flame_skin_gemv_kernel", 2.0), ("adam_flat_multi_kernel", 2.0)],
}


def main():
    fetch_db, write_db, tag, out = sys.argv[1:5]
    last = int(sys.argv[5]) if len(sys.argv) > 5 else 16
    f = summarise(fetch_db, last)
    w = summarise(write_db, last)
    res, raw = {}, {}
    for stage, kernels in STAGES.items():
        best = lo = hi = 0.0
        for kname, factor in kernels:
            for full in f:
                if kname + "E" in full or kname + "I" in full or kname + "(" in full or full.endswith(kname):
                    fb = f[full].get("FETCH_SIZE", 0.0) * 1024.0
                    wb = w.get(full, {}).get("WRITE_SIZE", 0.0) * 1024.0
                    best += factor * fb + wb
                    lo += fb + wb
                    hi += 2.0 * fb + wb
                    raw[full.split("(")[0][:60]] = {"fetch_bytes_counted": int(fb), "fetch_factor": factor, "write_bytes": int(wb)}
        res[f"{stage}@{tag}"] = int(best)
        res[f"{stage}@{tag}:raw"] = int(lo)
        res[f"{stage}@{tag}:fetch_x2"] = int(hi)
    D = None
    log = os.path.join(os.path.dirname(os.path.dirname(fetch_db)), os.path.basename(os.path.dirname(fetch_db)) + ".log")
    for cand in (log, fetch_db.replace("/pmc_results.db", ".log")):
        if os.path.exists(cand):
            m = re.findall(r"^D (\d+)", open(cand).read(), flags=re.M)
            if m:
                D = int(m[-1])
    res["_D"] = D
    res["_note"] = ("HBM bytes per launch from rocprofv3 --pmc (separate FETCH_SIZE and WRITE_SIZE passes of tools/run_scene.py --train, "
                    "averaged over the last %d launches).  `<stage>@<tag>` = factor * FETCH_SIZE * 1024 + WRITE_SIZE * 1024 with the factor of "
                    "profiles/r03_fetch_calibration.json per kernel (2 for coalesced streaming reads, 1 for 16- / 64-byte gathers: the "
                    "composite kernels); `:raw` and `:fetch_x2` are the two uniform readings.  `_D` = tile pairs of the profiled run." % last)
    res["_source"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `tools/run_scene.py --train --finetune` (eager launches), profile " + os.environ.get("OMFS_PROFILE_TAG", "?")
    json.dump(res, open(out, "w"), indent=1)
    json.dump(raw, open(out.replace(".json", "_raw.json"), "w"), indent=1)
    for k, v in res.items():
        if not k.startswith("_"):
            print(k, v)


if __name__ == "__main__":
    main()

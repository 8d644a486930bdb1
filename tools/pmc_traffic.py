#!/usr/bin/env python3
"""profiles/traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of tools/run_scene.py --train.

HBM bytes per launch = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024: the counters are in KiB, and FETCH_SIZE counts
half of what is read on gfx950 (MI355X_MICROARCH.md, HBM/rocprofv3 section; calibrated in round 1 on adam_kernel:
algorithmic reads 283 MB vs FETCH_SIZE 138 MB, writes 212 MB vs WRITE_SIZE 207 MB).  Kernels are grouped into the
stage names bench.py reports."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import summarise  # noqa: E402

STAGES = {
    "composite_fwd": ["composite_fwd_kernel", "composite_fwd_deep_kernel"],
    "composite_bwd": ["composite_bwd_kernel"],
    "loss": ["ssim_fwd_kernel", "ssim_bwd_kernel", "loss_reduce_kernel"],
    "tile_sort": ["tile_sort_kernel"],
    "bin_count": ["bin_count_kernel"],
    "bin_scan": ["tile_scan_kernel"],
    "bin_scatter": ["bin_scatter_kernel"],
    "project": ["project_fwd_kernel"],
    "project_bwd": ["project_bwd_kernel", "count_visible_kernel"],
    "adam": ["adam_kernel"],
    "flame": ["flame_joints_kernel", "flame_lbs_kernel", "face_frames_kernel"],
    "flame_bwd": ["face_frames_bwd_kernel", "flame_skin_bwd_kernel", "basis_t_gemv_kernel", "adam_flat_multi_kernel"],
}


def main():
    fetch_db, write_db, tag, out = sys.argv[1:5]
    last = int(sys.argv[5]) if len(sys.argv) > 5 else 16
    f = summarise(fetch_db, last)
    w = summarise(write_db, last)
    res, raw = {}, {}
    for stage, kernels in STAGES.items():
        total = 0.0
        for kname in kernels:
            for full in f:
                if kname in full:
                    fb = 2.0 * f[full].get("FETCH_SIZE", 0.0) * 1024.0
                    wb = w.get(full, {}).get("WRITE_SIZE", 0.0) * 1024.0
                    total += fb + wb
                    raw[full.split("(")[0][:60]] = {"fetch_bytes_corrected": int(fb), "write_bytes": int(wb)}
        res[f"{stage}@{tag}"] = int(total)
    res["_note"] = ("HBM bytes per launch from rocprofv3 --pmc (separate FETCH_SIZE and WRITE_SIZE passes, tools/run_scene.py "
                    "--train, averaged over the last %d launches): 2*FETCH_SIZE*1024 + WRITE_SIZE*1024; the x2 on FETCH_SIZE is "
                    "the gfx950 correction of MI355X_MICROARCH.md. Stages sum their kernels (tools/pmc_traffic.py)." % last)
    res["_source"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `tools/run_scene.py --train --finetune` (eager launches), profile " + os.environ.get("OMFS_PROFILE_TAG", "?")
    json.dump(res, open(out, "w"), indent=1)
    json.dump(raw, open(out.replace(".json", "_raw.json"), "w"), indent=1)
    for k, v in res.items():
        if not k.startswith("_"):
            print(k, v)


if __name__ == "__main__":
    main()

"""Study for DESIGN.md section 9 item 0b (no kernel is changed): if binning dropped, per tile, every pair deeper than a cut-off derived from
the view's PREVIOUS visit, how many pairs would be left, and how often would a cut list end before its pixels have saturated (an
iteration that would have to be redone with full lists)?  Runs the ordinary trainer on the bench scene and evaluates the
hypothetical cut on the full lists of every visit.

  cut_t(view) = depth of the entry at position min(len - 1, a * depth_pos + b) of tile t's list at the previous visit, where depth_pos is
                the deepest last contributor of the tile's pixels; +inf when a pixel of the tile was still unsaturated at the end
                of its list (final_T above 1e-2: the stop rule leaves at most that) or the tile had no list
  at the next visit: kept_t = #{entries of tile t with depth <= cut_t}; VIOLATION when the deepest last contributor of the visit lies
                behind kept_t, or a pixel is unsaturated at the end of a list that was cut.

usage (GPU box): python tools/depth_cut_study.py [--iters 400] [--start 0]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omfs_4d_video_gen_amd.engine import synthetic  # noqa: E402
from omfs_4d_video_gen_amd.engine.flame_rig import FlameRig  # noqa: E402
from omfs_4d_video_gen_amd.engine.trainer import Renderer, Trainer, View  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=400)
a = ap.parse_args()
N, W, H = 300000, 1920, 1080
rig = FlameRig.from_synthetic(synthetic.make_rig(0))
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
r = t.rast
gy, gx, nt = r.gy, r.gx, r.n_tiles
RULES = [(1.25, 32), (1.5, 64), (2.0, 64), (2.0, 128), (3.0, 128)]


def tile_stats():
    """per tile: list length, deepest last contributor, a pixel unsaturated at the end, and the lists' depths (sorted)"""
    ts = r.tile_start.to(torch.int64)
    D = int(ts[-1])
    ids = r.sorted_ids[:D].to(torch.int64)
    depth = r.g2[:, 1].contiguous()[ids]                                   # ascending inside every tile
    pad_n = torch.zeros(gy * 16, gx * 16, dtype=torch.int64, device="cuda")
    pad_n[:H, :W] = r.n_contrib.view(H, W).to(torch.int64)
    pad_T = torch.zeros(gy * 16, gx * 16, device="cuda")
    pad_T[:H, :W] = r.final_T.view(H, W)
    tile = lambda x: x.view(gy, 16, gx, 16).permute(0, 2, 1, 3).reshape(nt, 256)
    pos = tile(pad_n).max(1).values
    live = (tile(pad_T) > 1e-2).any(1)
    return ts, depth, pos, live


prev = {}            # view index -> {rule: cut depth per tile}
tot = {rule: {"visits": 0, "violating_visits": 0, "violating_tiles": 0, "kept": 0, "full": 0, "visited": 0} for rule in RULES}
first_visits = 0
for it in range(a.iters):
    from omfs_4d_video_gen_amd.engine.distributed import view_index
    vi = view_index(t.step_idx, 0, 1, len(views), t.view_seed)
    t.step()
    torch.cuda.synchronize()
    ts, depth, pos, live = tile_stats()
    length = ts[1:] - ts[:-1]
    D = int(ts[-1])
    tile_of = torch.repeat_interleave(torch.arange(nt, device="cuda"), length)
    if vi in prev:
        for rule in RULES:
            cut = prev[vi][rule]
            keep = depth <= cut[tile_of]
            kept = torch.zeros(nt, dtype=torch.int64, device="cuda").index_add_(0, tile_of, keep.to(torch.int64))
            was_cut = kept < length
            viol = was_cut & ((pos > kept) | live)
            s = tot[rule]
            s["visits"] += 1
            s["violating_visits"] += int(viol.any())
            s["violating_tiles"] += int(viol.sum())
            s["kept"] += int(kept.sum()); s["full"] += D; s["visited"] += int(pos.sum())
    else:
        first_visits += 1
    cuts = {}
    for (ka, kb) in RULES:
        p = (ka * pos.to(torch.float64)).to(torch.int64) + kb
        p = torch.minimum(p, torch.clamp(length - 1, min=0))
        idx = torch.clamp(ts[:-1] + p, max=max(D - 1, 0))
        c = depth[idx] if D > 0 else torch.zeros(nt, device="cuda")
        nocut = live | (length == 0) | (p >= length - 1)
        cuts[(ka, kb)] = torch.where(nocut, torch.full_like(c, float("inf")), c)
    prev[vi] = cuts
out = {"iterations": a.iters, "first_visits_without_a_hint": first_visits, "rules": {}}
for rule, s in tot.items():
    if s["visits"]:
        out["rules"][f"{rule[0]} x depth + {rule[1]}"] = {
            "hinted_visits": s["visits"], "visits_with_a_violation": s["violating_visits"], "violating_tiles_per_visit": round(s["violating_tiles"] / s["visits"], 2),
            "pairs_kept_of_full": round(s["kept"] / max(s["full"], 1), 3), "pairs_visited_of_full": round(s["visited"] / max(s["full"], 1), 3)}
print(json.dumps(out, indent=1))

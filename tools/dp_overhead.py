"""Fixed cost of the data-parallel exchange path on ONE GPU: the bench workload (FLAME fine-tuning on, as the headline) stepped
with the exchange forced over a one-rank RCCL group (OMFS_DP_FORCE=1) -- compact, full, sharded -- against the plain step.
What is measured is everything but the link time: the collectives' launches, their stream hand-overs, the rebuilt SH
gradients and the split Adam; and, beside the GPU time per step, the HOST's enqueue time per step (the Python loop's own
wall clock before the final sync): the exchange path must not become host bound at 8 ranks.
usage (GPU box): python tools/dp_overhead.py [--steps 200] [--out profiles/r04_dp_overhead.json]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(mode, steps):
    import torch
    import torch.distributed as dist
    from omfs_4d_video_gen_amd.engine import synthetic
    from omfs_4d_video_gen_amd.engine.flame_rig import FlameRig
    from omfs_4d_video_gen_amd.engine.trainer import Renderer, Trainer, View
    N, W, H, V = 300000, 1920, 1080, 16
    srig = synthetic.make_rig(0)
    rig = FlameRig.from_synthetic(srig)
    seq = synthetic.make_flame_sequence(V, 0)
    cams = synthetic.make_camera_arc(W, H, V)
    tr = Renderer(rig, seq, synthetic.make_gaussians(N, rig.n_faces, 1), W, H)
    views = []
    for i, cam in enumerate(cams):
        v = View(cam, timestep=i % V)
        v.target = tr.render(v).clone()
        views.append(v)
    del tr
    pg = None
    os.environ["OMFS_SH_ADAM"] = "0" if mode == "plain_all_planes" else "1"
    os.environ["OMFS_DP_IMPL"] = "abi" if mode.endswith("-abi") else "torch"
    if not mode.startswith("plain"):
        os.environ["OMFS_DP_FORCE"] = "1"
        os.environ["OMFS_DP_EXCHANGE"] = "full" if mode.startswith("full") else ("compact" if mode.startswith("compact") else mode)
        pg = dist.group.WORLD
    else:
        os.environ.pop("OMFS_DP_FORCE", None)
    t = Trainer(rig, seq, synthetic.make_gaussians(N, rig.n_faces, 0), views, W, H, iterations=30000, start_sh_degree=3,
                rank=0, world_size=1, process_group=pg, finetune_flame=True)
    for _ in range(20):
        t.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        t.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    del t
    return {"ms_per_step": round((t2 - t0) / steps * 1e3, 4), "host_enqueue_ms_per_step": round((t1 - t0) / steps * 1e3, 4)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    # plain = the single-GPU step as benchmarked (45 SH gradient planes formed inside the Adam launch); plain_all_planes = the
    # same with all 59 planes through the gradient buffer, which is what every exchange mode starts from; full-abi = the full
    # exchange issued by the library's own communicator (omfs_rccl_allreduce_grads) instead of torch.distributed
    out = {m: run(m, a.steps) for m in ("plain", "plain_all_planes", "compact", "compact-abi", "full", "full-abi", "sharded")}
    base = out["plain_all_planes"]["ms_per_step"]
    rec = {"workload": "bench default (300k Gaussians, 1920x1080, 16 views, FLAME fine-tuning on)", "steps": a.steps, "modes": out,
           "overhead_us_over_plain_all_planes": {m: round((v["ms_per_step"] - base) * 1e3, 1) for m, v in out.items() if not m.startswith("plain")},
           "note": "one-rank RCCL group (OMFS_DP_FORCE=1): the exchange path without link time; UNMEASURED on more than one GPU"}
    print(json.dumps(rec))
    if a.out:
        with open(a.out, "w") as f:
            json.dump(rec, f, indent=1)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Where the HIP backward and the autograd oracle differ most (per parameter group): the worst Gaussians, their projected
quantities, and the same gradient from a float64 run of the oracle -- tells rounding of the fp32 ORACLE from an error of the
engine.  usage: python tools/diag_backward.py [n width height]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H  # noqa: E402
from omfs_4d_video_gen_amd.engine import synthetic  # noqa: E402
from omfs_4d_video_gen_amd.engine.flame_rig import FlameRig, DeviceFlame  # noqa: E402
from omfs_4d_video_gen_amd.engine.gaussians import GaussianModel, pack_params  # noqa: E402
from omfs_4d_video_gen_amd.engine.rasterizer import Rasterizer, make_camera_struct  # noqa: E402
from oracle import c_oracle as CO  # noqa: E402
from oracle import torch_splat as O  # noqa: E402


def main():
    n, width, height = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (40000, 448, 252)
    seed, bg, t = 4, (0.0, 0.0, 0.0), 1
    rig = synthetic.make_rig(seed)
    g = synthetic.make_gaussians(n, rig.faces.shape[0], seed)
    seq = synthetic.make_flame_sequence(3, seed)
    cam = synthetic.make_camera(width, height, yaw=0.25)
    dflame = DeviceFlame(FlameRig.from_synthetic(rig), seq)
    model, rast = GaussianModel(g), Rasterizer(n, width, height)
    _, face_xf = dflame.face_frames(t, 1)
    ccam = make_camera_struct(cam, sh_degree=3, bg=bg)
    rast.forward(model, face_xf[0], ccam)
    dimage = torch.randn(3, height, width, generator=torch.Generator().manual_seed(7))
    grads = torch.zeros(59, model.n_pad, device="cuda")
    reg = (0.01, 1.0, 1.0, 0.6)
    rast.backward(model, face_xf[0], ccam, grads, dimage=dimage.cuda().contiguous(), reg=reg)
    torch.cuda.synchronize()
    cref = CO.render(dflame, t, pack_params(g), g["binding"], n, CO.camera(ccam))
    lists = O.lists_from_offsets(cref["tile_start"], cref["ids"])
    gh = grads[:, :n].cpu().numpy().astype(np.float64)
    got = {"xyz": gh[0:3].T, "log_scale": gh[3:6].T, "rot": gh[6:10].T, "opacity": gh[10], "sh": gh[11:].T.reshape(n, 16, 3)}
    res = {}
    for dt in (torch.float32, torch.float64):
        torch.set_default_dtype(dt)
        og = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in H.oracle_gaussians(g).items()}
        for k in ("xyz", "log_scale", "rot", "opacity", "sh"):
            og[k].requires_grad_(True)
        orig = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in H.oracle_rig(rig).items()}
        fr = {k: (v.to(dt) if v is not None and v.is_floating_point() else v) for k, v in H.oracle_frame(seq, t).items()}
        ref = O.render(orig, og, fr, cam, bg=bg, sh_degree=3, lists=lists)
        loss = (ref["image"] * dimage.to(dt)).sum() + O.regularisers(og, ref["proj"]["visible"], *reg)
        loss.backward()
        res[dt] = ({k: og[k].grad.numpy().astype(np.float64) for k in got}, ref)
    torch.set_default_dtype(torch.float32)
    r32, r64 = res[torch.float32][0], res[torch.float64][0]
    proj = res[torch.float64][1]["proj"]
    for name in got:
        scale = np.abs(r64[name]).max()
        d_e64 = np.abs(got[name] - r64[name]).reshape(n, -1).max(1)
        d_e32 = np.abs(got[name] - r32[name]).reshape(n, -1).max(1)
        d_3264 = np.abs(r32[name] - r64[name]).reshape(n, -1).max(1)
        print(f"{name}: max|ref64| {scale:.4g}  engine-vs-64 {d_e64.max() / scale:.2e}  engine-vs-32 {d_e32.max() / scale:.2e}  "
              f"oracle32-vs-64 {d_3264.max() / scale:.2e}   (relative to the group maximum)")
        for i in np.argsort(-d_e64)[:3]:
            print(f"   id {i}: engine-64 {d_e64[i]:.3e} engine-32 {d_e32[i]:.3e} 32-64 {d_3264[i]:.3e} | opac {float(proj['opac'][i]):.4f} "
                  f"radius {int(proj['radius'][i])} mean2d {[round(float(x), 2) for x in proj['mean2d'][i]]} "
                  f"conic {[round(float(x), 4) for x in proj['conic'][i]]} depth {float(proj['depth'][i]):.4f}")


if __name__ == "__main__":
    main()

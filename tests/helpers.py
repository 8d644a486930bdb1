"""Shared builders for tests: same seeded inputs for the HIP path and the oracle."""
import numpy as np
import torch

from omfs_4d_video_gen_amd.engine import synthetic
from omfs_4d_video_gen_amd.engine.flame_rig import FlameRig, pose_rotmats


def oracle_rig(rig):
    """synthetic rig -> dict of torch tensors for oracle.torch_splat."""
    return {"v_template": torch.from_numpy(rig.v_template), "shapedirs": torch.from_numpy(rig.shapedirs),
            "posedirs": torch.from_numpy(rig.posedirs), "J_regressor": torch.from_numpy(rig.J_regressor),
            "weights": torch.from_numpy(rig.weights), "faces": torch.from_numpy(rig.faces.astype(np.int64))}


def oracle_frame(seq, t):
    rm = pose_rotmats(seq)
    return {"shape": torch.from_numpy(seq["shape"]), "expr": torch.from_numpy(seq["expr"][t]),
            "rotmats": torch.from_numpy(rm[t]), "translation": torch.from_numpy(seq["translation"][t]),
            "static_offset": torch.from_numpy(seq["static_offset"][0]),
            "dynamic_offset": torch.from_numpy(seq["dynamic_offset"][t])}


def oracle_gaussians(g, requires_grad=False):
    d = {k: torch.from_numpy(np.asarray(v)).clone() for k, v in g.items()}
    if requires_grad:
        for k in ("xyz", "log_scale", "rot", "opacity", "sh"):
            d[k].requires_grad_(True)
    return d


def check_backward_on_tile_subset(N, W, H, yaw, seed, t=2, n_heavy=16, n_other=48, min_heavy_len=512, bg=(0.05, 0.1, 0.15)):
    """HIP backward against the autograd oracle at sizes where the oracle cannot composite the whole image: dL/dimage is
    non-zero on n_heavy + n_other fixed tiles only (the heaviest lists -- many segments, deep forward -- and seeded others),
    so the engine's whole forward + backward runs at full size while the oracle composites just those tiles (with the
    engine's = the C oracle's lists: one order on both sides) and differentiates through its own projection of all N
    Gaussians.  Per element 2e-3 of the group maximum, 2e-4 in aggregate (as tests/test_gpu_backward.py)."""
    from omfs_4d_video_gen_amd.engine.flame_rig import DeviceFlame
    from omfs_4d_video_gen_amd.engine.gaussians import GaussianModel
    from omfs_4d_video_gen_amd.engine.rasterizer import Rasterizer, make_camera_struct
    from oracle import torch_splat as O
    rig = synthetic.make_rig(seed)
    g = synthetic.make_gaussians(N, rig.faces.shape[0], seed)
    seq = synthetic.make_flame_sequence(4, seed)
    cam = synthetic.make_camera(W, H, yaw=yaw)
    dflame = DeviceFlame(FlameRig.from_synthetic(rig), seq)
    _, face_xf = dflame.face_frames(t, 1)
    ccam = make_camera_struct(cam, sh_degree=3, bg=bg)
    model, rast = GaussianModel(g), Rasterizer(N, W, H)
    rast.forward(model, face_xf[0], ccam)
    torch.cuda.synchronize()
    rast.check_status()
    ts = rast.tile_start.cpu().numpy().astype(np.int64)
    ids = rast.sorted_ids.cpu().numpy().view(np.uint32)[:ts[-1]]
    lens = np.diff(ts)
    heavy = np.argsort(-lens)[:n_heavy]
    rest = np.setdiff1d(np.nonzero(lens > 0)[0], heavy)
    tiles = sorted(set(heavy.tolist()) | set(np.random.default_rng(11).choice(rest, n_other, replace=False).tolist()))
    assert len(tiles) == n_heavy + n_other and int(lens[heavy].min()) > min_heavy_len
    gx = rast.gx
    mask = torch.zeros(H, W)
    for tl in tiles:
        ty, tx = divmod(tl, gx)
        mask[ty * 16:(ty + 1) * 16, tx * 16:(tx + 1) * 16] = 1.0
    dimage = torch.randn(3, H, W, generator=torch.Generator().manual_seed(7)) * mask
    grads = torch.zeros(59, model.n_pad, device="cuda")
    reg = (0.01, 1.0, 1.0, 0.6)
    rast.backward(model, face_xf[0], ccam, grads, dimage=dimage.cuda().contiguous(), reg=reg)
    torch.cuda.synchronize()

    torch.set_num_threads(8)
    og = oracle_gaussians(g, requires_grad=True)
    ref = O.render(oracle_rig(rig), og, oracle_frame(seq, t), cam, bg=bg, sh_degree=3,
                   lists=O.lists_from_offsets(ts, ids), tiles=set(tiles))
    img = rast.image.cpu()
    on = mask.bool()
    assert float((img[:, on] - ref["image"][:, on]).abs().mean()) < 1e-4
    loss = (ref["image"] * dimage).sum() + O.regularisers(og, ref["proj"]["visible"], *reg)
    loss.backward()
    gh = grads[:, :N].cpu().numpy()
    got = {"xyz": gh[0:3].T, "log_scale": gh[3:6].T, "rot": gh[6:10].T, "opacity": gh[10], "sh": gh[11:].T.reshape(N, 16, 3)}
    touched = np.unique(np.concatenate([ids[ts[tl]:ts[tl + 1]] for tl in tiles]))
    untouched = np.ones(N, bool)
    untouched[touched] = False
    for name, gt in got.items():
        r = og[name].grad.numpy()
        scale = np.abs(r).max()
        d = np.abs(gt - r)
        assert d.max() <= 2e-3 * scale + 1e-7, f"{name}: max diff {d.max()} vs max ref {scale}"
        assert d.sum() <= 2e-4 * np.abs(r).sum() + 1e-6, name
        # Gaussians outside the chosen tiles receive the regularisers' gradient only
        assert np.abs(gt[untouched] - r[untouched]).max() <= 1e-6 * max(scale, 1.0), name
    return rast, tiles

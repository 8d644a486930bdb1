"""GPU: gradients of the FLAME parameters (expression, joint poses, translation) through
project_bwd -> omfs_face_frames_bwd -> omfs_flame_skin_bwd -> omfs_flame_param_bwd (engine/flame_finetune.py),
against PyTorch-CPU autograd through the oracle's full FLAME + splat forward; device rodrigues vs the pinned host one."""
import numpy as np
import pytest
import torch

import helpers as H
from omfs_4d_video_gen_amd.engine import synthetic

pytestmark = pytest.mark.gpu


def _setup(n, width, height, seed=3):
    from omfs_4d_video_gen_amd.engine.flame_rig import FlameRig, DeviceFlame
    from omfs_4d_video_gen_amd.engine.flame_finetune import FlameFineTuner
    from omfs_4d_video_gen_amd.engine.gaussians import GaussianModel
    from omfs_4d_video_gen_amd.engine.rasterizer import Rasterizer, make_camera_struct
    rig = synthetic.make_rig(seed)
    g = synthetic.make_gaussians(n, rig.faces.shape[0], seed)
    seq = synthetic.make_flame_sequence(4, seed)
    cam = synthetic.make_camera(width, height, yaw=0.3)
    dflame = DeviceFlame(FlameRig.from_synthetic(rig), seq)
    ft = FlameFineTuner(dflame, seq)
    return rig, g, seq, cam, dflame, ft, GaussianModel(g), Rasterizer(n, width, height), make_camera_struct


@pytest.mark.parametrize("n,width,height", [(1500, 96, 80), (6000, 160, 128)])
def test_flame_parameter_gradients_match_autograd(n, width, height):
    from oracle import torch_splat as O
    rig, g, seq, cam, dflame, ft, model, rast, mk = _setup(n, width, height)
    t = 2
    bg = (0.2, 0.1, 0.3)
    ccam = mk(cam, sh_degree=3, bg=bg)
    ft.begin(t, model.binding)
    verts, face_xf = dflame.face_frames(t, 1)
    rast.forward(model, face_xf[0], ccam)
    gen = torch.Generator().manual_seed(5)
    dimage = torch.randn(3, height, width, generator=gen)
    grads = torch.zeros(59, model.n_pad, device="cuda")
    rast.backward(model, face_xf[0], ccam, grads, dimage=dimage.cuda().contiguous(), reg=(0.0, 1.0, 0.0, 0.6), dface=ft.dface)
    ft.backward(verts[0])
    torch.cuda.synchronize()
    got = {"expr": ft.grad["expr"][t].cpu(), "pose": ft.grad["pose"][t].cpu().reshape(5, 3), "translation": ft.grad["translation"][t].cpu()}
    # rows of the other timesteps stay untouched
    others = [i for i in range(ft.expr.shape[0]) if i != t]
    assert float(ft.grad["expr"][others].abs().max()) == 0.0 and float(ft.grad["pose"][others].abs().max()) == 0.0

    # oracle: the same parameters as leaves of the full differentiable forward
    expr = torch.from_numpy(seq["expr"][t]).clone().requires_grad_(True)
    eyes = np.asarray(seq.get("eyes_pose", np.zeros((4, 6), np.float32)), np.float32).reshape(-1, 6)
    pose_np = np.stack([np.asarray(seq["rotation"], np.float32).reshape(-1, 3)[t],
                        np.asarray(seq.get("neck_pose", np.zeros((4, 3))), np.float32).reshape(-1, 3)[t],
                        np.asarray(seq.get("jaw_pose", np.zeros((4, 3))), np.float32).reshape(-1, 3)[t], eyes[t, :3], eyes[t, 3:]])
    pose = torch.from_numpy(pose_np).clone().requires_grad_(True)
    trans = torch.from_numpy(np.asarray(seq["translation"], np.float32).reshape(-1, 3)[t]).clone().requires_grad_(True)
    frame = H.oracle_frame(seq, t)
    frame["expr"], frame["rotmats"], frame["translation"] = expr, O.rodrigues(pose), trans
    ref = O.render(H.oracle_rig(rig), H.oracle_gaussians(g), frame, cam, bg=bg, sh_degree=3)
    (ref["image"] * dimage).sum().backward()
    for name, r in (("expr", expr.grad), ("pose", pose.grad), ("translation", trans.grad)):
        d = float((got[name] - r).abs().max())
        scale = float(r.abs().max())
        assert scale > 0
        assert d <= 5e-3 * scale + 1e-6, f"{name}: max diff {d} vs max ref {scale}"


def test_fused_flame_launches_equal_the_split_ones(monkeypatch):
    """ABI 6: omfs_flame_pose_lbs (joints + skinning, one frame) writes the SAME BITS as omfs_flame_joints_pose + omfs_flame_lbs,
    and omfs_flame_skin_param_bwd the gradients of omfs_flame_skin_bwd + omfs_flame_param_bwd (sums reordered: float noise);
    a second call of the fused backward finds its accumulators zeroed by the first."""
    rig, g, seq, cam, dflame, ft, model, rast, mk = _setup(3000, 128, 96)
    t = 1
    ccam = mk(cam, sh_degree=3, bg=(0.0, 0.0, 0.0))

    def run(split):
        monkeypatch.setenv("OMFS_FLAME_SPLIT", "1" if split else "0")
        ft.fused = not split
        out = []
        for _ in range(2):
            ft.begin(t, model.binding)
            verts, face_xf = dflame.face_frames(t, 1)
            joint_xf, coef, _, _, v_shaped = dflame._buffers(1)
            rast.forward(model, face_xf[0], ccam)
            dimage = torch.randn(3, 96, 128, generator=torch.Generator().manual_seed(11)).cuda()
            grads = torch.zeros(59, model.n_pad, device="cuda")
            rast.backward(model, face_xf[0], ccam, grads, dimage=dimage, reg=(0.0, 1.0, 0.0, 0.6), dface=ft.dface)
            ft.backward(verts[0])
            torch.cuda.synchronize()
            out.append({"verts": verts.clone(), "face_xf": face_xf.clone(), "joint_xf": joint_xf.clone(), "coef": coef[:, 0].clone(),
                        "v_shaped": v_shaped.clone(), "rotmats": dflame.rotmats[t].clone(),
                        "g": torch.cat([ft.grad["expr"][t].flatten(), ft.grad["pose"][t].flatten(), ft.grad["translation"][t].flatten()]).clone()})
            ft.grad_flat.zero_()
        return out
    split = run(True)
    ft.sums.zero_(); ft.dcoef.zero_()       # the split launches overwrite their scratch; the fused one accumulates into zeroed words
    fused = run(False)
    for k in ("verts", "face_xf", "joint_xf", "coef", "v_shaped", "rotmats"):
        assert torch.equal(fused[0][k], split[0][k]), k
    scale = float(split[0]["g"].abs().max())
    assert scale > 0
    for a in fused:
        assert float((a["g"] - split[0]["g"]).abs().max()) <= 2e-5 * scale
    assert float(ft.dcoef.abs().max()) == 0.0 and float(ft.sums.abs().max()) == 0.0 and float(ft.dverts.abs().max()) == 0.0


def test_device_rodrigues_matches_the_pinned_host_formula():
    from omfs_4d_video_gen_amd import _lib as L
    from omfs_4d_video_gen_amd.engine.flame_rig import rodrigues
    gen = torch.Generator().manual_seed(1)
    aa = torch.randn(257, 3, generator=gen) * torch.logspace(-6, 0.5, 257)[:, None]
    aa[0] = 0.0
    out = torch.empty(257, 9, device="cuda")
    L.check(L.load().omfs_flame_rodrigues(L.ptr(aa.cuda().contiguous()), 257, L.ptr(out), L.stream_ptr()), "omfs_flame_rodrigues")
    assert torch.allclose(out.cpu().reshape(-1, 3, 3), rodrigues(aa), atol=2e-6)


def test_finetuning_moves_a_perturbed_pose_back():
    """Train only the FLAME parameters against targets rendered with the true sequence: a perturbed jaw / translation
    must move towards the truth (the photometric loss decreases)."""
    from omfs_4d_video_gen_amd.engine.flame_rig import FlameRig
    from omfs_4d_video_gen_amd.engine.trainer import Renderer, Trainer, View
    n, W, Hh = 8000, 192, 160
    srig = synthetic.make_rig(0)
    rig = FlameRig.from_synthetic(srig)
    seq = synthetic.make_flame_sequence(4, 0)
    cams = synthetic.make_camera_arc(W, Hh, 4)
    g = synthetic.make_gaussians(n, rig.n_faces, 0)
    rr = Renderer(rig, seq, g, W, Hh)
    views = []
    for i, c in enumerate(cams):
        v = View(c, timestep=i)
        v.target = rr.render(v).clone()
        views.append(v)
    bad = {k: np.array(v) for k, v in seq.items()}
    bad["translation"] = np.asarray(bad["translation"], np.float32).reshape(-1, 3) + np.array([0.004, -0.003, 0.0], np.float32)
    tr = Trainer(rig, bad, g, views, W, Hh, start_sh_degree=3, finetune_flame=True,
                 flame_lr={"translation": 2e-4, "expr": 1e-3, "pose": 1e-4})
    tr.opt.set_lr(np.zeros(59, np.float32))     # Gaussians frozen: only the FLAME parameters may explain the images
    tr.pos_lr = (0.0, 0.0)
    losses = []
    for it in range(120):
        tr.step()
        if it % 4 == 0:
            losses.append(tr.loss_value())
    torch.cuda.synchronize()        # the FLAME parameters are updated on the trainer's side stream
    t_err0 = 0.005
    t_now = tr.flame_ft.translation.cpu().numpy() - np.asarray(seq["translation"], np.float32).reshape(-1, 3)
    assert np.mean(losses[-5:]) < 0.7 * np.mean(losses[:5]), (losses[:5], losses[-5:])
    assert float(np.linalg.norm(t_now, axis=1).mean()) < 0.7 * t_err0


def test_multi_tensor_adam_stays_inside_its_tensors_and_matches_torch():
    """omfs_adam_flat_multi: three tensors whose sizes are not multiples of the block size, laid out back to back between
    guard words (a thread of a segment's last block that owns nothing must not touch the NEXT segment's memory at a
    negative index); the update equals torch.optim.Adam's, the gradients are consumed."""
    import ctypes as C
    from omfs_4d_video_gen_amd import _lib as L
    sizes, lrs = (600, 90, 18), (1e-3, 1e-5, 1e-6)
    guard = 512
    gen = torch.Generator().manual_seed(3)
    bufs = {}
    for role in "pgmv":          # each role: [guard | t0 | guard | t1 | guard | t2 | guard] in ONE allocation
        total = guard + sum(n + guard for n in sizes)
        bufs[role] = torch.full((total,), 7.0, device="cuda")
    def views(role):
        out, o = [], guard
        for n in sizes:
            out.append(bufs[role][o:o + n]); o += n + guard
        return out
    P, G, M, V = views("p"), views("g"), views("m"), views("v")
    ref_p = []
    for k, n in enumerate(sizes):
        P[k].copy_(torch.randn(n, generator=gen)); M[k].zero_(); V[k].zero_()
        ref_p.append(P[k].cpu().clone().requires_grad_(True))
    opt = torch.optim.Adam([{"params": [ref_p[k]], "lr": lrs[k]} for k in range(3)], eps=1e-15)
    arr = lambda ts: (C.c_void_p * 3)(*[L.ptr(t) for t in ts])
    for step in range(1, 4):
        for k, n in enumerate(sizes):
            gk = torch.randn(n, generator=gen) * 0.1
            G[k].copy_(gk); ref_p[k].grad = gk.clone()
        L.check(L.load().omfs_adam_flat_multi(3, arr(P), arr(G), arr(M), arr(V), (C.c_int * 3)(*sizes), (C.c_float * 3)(*lrs),
                                              0.9, 0.999, 1e-15, step, 1.0, 0, L.stream_ptr()), "omfs_adam_flat_multi")
        opt.step()
        torch.cuda.synchronize()
        for k in range(3):
            assert float(G[k].abs().max()) == 0.0                                  # consumed
            assert torch.allclose(P[k].cpu(), ref_p[k].detach(), rtol=2e-5, atol=1e-7), (step, k)
    for role in "pgmv":          # every guard word is still the fill value
        mask = torch.ones_like(bufs[role], dtype=torch.bool)
        o = guard
        for n in sizes:
            mask[o:o + n] = False; o += n + guard
        assert bool((bufs[role][mask] == 7.0).all()), role

"""GPU parity of the backward path against PyTorch-CPU autograd through the oracle:
composite backward + projection/deformation backward (+ regularisers), the L1+D-SSIM loss and
the fused Adam.  fp32 tolerance: per parameter group, max|diff| <= 2e-3 * max|ref| + 1e-7
(float atomics reorder sums; the oracle differentiates through torch's own exp/matmul), see helpers.assert_grads_close."""
import os

import numpy as np
import pytest
import torch

import helpers as H
from omfs_4d_video_gen_amd.engine import synthetic

pytestmark = pytest.mark.gpu

GROUPS = {"xyz": (0, 3), "log_scale": (3, 6), "rot": (6, 10), "opacity": (10, 11), "sh": (11, 59)}


def _scene(n, width, height, seed=0, big_scale=False, identity=False):
    from omfs_4d_video_gen_amd.engine.flame_rig import FlameRig, DeviceFlame
    from omfs_4d_video_gen_amd.engine.gaussians import GaussianModel
    from omfs_4d_video_gen_amd.engine.rasterizer import Rasterizer, make_camera_struct
    rig = synthetic.make_rig(seed)
    g = synthetic.make_gaussians(n, rig.faces.shape[0], seed)
    if big_scale:   # push some Gaussians over the regulariser thresholds
        g["log_scale"][: n // 4] += 1.0
        g["xyz"][: n // 8] *= 6.0
    seq = synthetic.make_flame_sequence(3, seed, identity=identity)
    cam = synthetic.make_camera(width, height, yaw=0.0 if identity else 0.25)
    dflame = DeviceFlame(FlameRig.from_synthetic(rig), seq)
    return rig, g, seq, cam, dflame, GaussianModel(g), Rasterizer(n, width, height), make_camera_struct


def _grads_to_groups(grads, n):
    gh = grads[:, :n].cpu().numpy()
    return {"xyz": gh[0:3].T, "log_scale": gh[3:6].T, "rot": gh[6:10].T, "opacity": gh[10],
            "sh": gh[11:].T.reshape(n, 16, 3)}


@pytest.mark.parametrize("n,width,height,bg,big", [(1200, 80, 64, (1.0, 1.0, 1.0), False), (3000, 128, 96, (0.1, 0.2, 0.3), True),
                                                   (16000, 64, 48, (0.3, 0.1, 0.2), False),    # deep lists
                                                   (5000, 256, 256, (0.0, 0.0, 0.0), False),   # BASELINE config 1 at its stated size (identity pose)
                                                   (40000, 448, 252, (0.0, 0.0, 0.0), False)])   # as large as autograd takes in a few seconds
def test_backward_matches_autograd(n, width, height, bg, big):
    from oracle import torch_splat as O
    config1 = (n, width, height) == (5000, 256, 256)
    rig, g, seq, cam, dflame, model, rast, mk = _scene(n, width, height, seed=0 if config1 else 4, big_scale=big, identity=config1)
    t = 1
    _, face_xf = dflame.face_frames(t, 1)
    ccam = mk(cam, sh_degree=3, bg=bg)
    rast.forward(model, face_xf[0], ccam)
    gen = torch.Generator().manual_seed(7)
    dimage = torch.randn(3, height, width, generator=gen)
    grads = torch.zeros(59, model.n_pad, device="cuda")
    reg = (0.01, 1.0, 1.0, 0.6)
    rast.backward(model, face_xf[0], ccam, grads, dimage=dimage.cuda().contiguous(), reg=reg)
    torch.cuda.synchronize()

    # both oracles composite in ONE order: the per-tile lists come from the bit-level spec (oracle/splat_oracle.c), whose
    # lists the engine reproduces bit for bit (test_gpu_bitexact.py) -- asserted here again, since everything below rests on it
    from omfs_4d_video_gen_amd.engine.gaussians import pack_params
    from oracle import c_oracle as CO
    cref = CO.render(dflame, t, pack_params(g), g["binding"], n, CO.camera(ccam))
    D = int(rast.tile_start[-1])
    assert np.array_equal(rast.tile_start.cpu().numpy().view(np.uint32), cref["tile_start"])
    assert np.array_equal(rast.sorted_ids.cpu().numpy().view(np.uint32)[:D], cref["ids"])
    og = H.oracle_gaussians(g, requires_grad=True)
    diag = {}
    ref = O.render(H.oracle_rig(rig), og, H.oracle_frame(seq, t), cam, bg=bg, sh_degree=3,
                   lists=O.lists_from_offsets(cref["tile_start"], cref["ids"]), decide=cref["proj"], diag=diag)
    loss = (ref["image"] * dimage).sum() + O.regularisers(og, ref["proj"]["visible"], *reg)
    loss.backward()
    got = _grads_to_groups(grads, n)
    assert int(rast.n_visible.item()) == int(ref["proj"]["visible"].sum())
    H.assert_grads_close(got, {k: og[k].grad.numpy() for k in GROUPS}, diag["near_gaussians"], n)


@pytest.mark.parametrize("W,Hh", [(75, 50), (200, 121), (64, 34)])
def test_loss_l1_ssim_value_and_gradient(W, Hh):
    # sizes that are not multiples of the 64-column x 34-row strips exercise the zero-padded borders
    from oracle import torch_splat as O
    from omfs_4d_video_gen_amd.engine.rasterizer import Rasterizer
    rast = Rasterizer(256, W, Hh)
    gen = torch.Generator().manual_seed(3)
    img = torch.rand(3, Hh, W, generator=gen)
    gt = (img + 0.2 * torch.randn(3, Hh, W, generator=gen)).clamp(0, 1)
    rast.image.copy_(img)
    rast._ensure_bwd()
    rast.loss.zero_()
    rast.loss_l1_ssim(gt.cuda().contiguous(), 0.2)
    torch.cuda.synchronize()
    x = img.clone().requires_grad_(True)
    ref = O.photometric_loss(x, gt, 0.2)
    ref.backward()
    assert abs(float(rast.loss.item()) - float(ref)) < 2e-6 * max(1.0, abs(float(ref)))
    d = (rast.dimage.cpu() - x.grad).abs().max().item()
    assert d < 1e-7 + 1e-3 * x.grad.abs().max().item(), d


def test_adam_matches_torch_adam():
    from omfs_4d_video_gen_amd.engine.gaussians import GaussianModel
    from omfs_4d_video_gen_amd.engine.rasterizer import Adam, default_lr_planes
    rig = synthetic.make_rig(0)
    n = 1000
    g = synthetic.make_gaussians(n, rig.faces.shape[0], 0)
    model = GaussianModel(g)
    lr = default_lr_planes()
    opt = Adam(model, lr)
    p_ref = model.params.cpu().clone()
    plist = [p_ref[i].clone().requires_grad_(True) for i in range(59)]
    topt = torch.optim.Adam([{"params": [plist[i]], "lr": float(lr[i])} for i in range(59)], eps=1e-15)
    gen = torch.Generator().manual_seed(1)
    for step in range(5):
        grads = torch.randn(59, model.n_pad, generator=gen) * 0.01
        grads[:, n:] = 0
        if step == 2:
            grads[:, : n // 2] = 0      # untouched Gaussians still decay their moments
        opt.step(grads.cuda().contiguous())
        for i in range(59):
            plist[i].grad = grads[i].clone()
        topt.step()
    torch.cuda.synchronize()
    got = model.params.cpu()
    ref = torch.stack([p.detach() for p in plist])
    assert torch.allclose(got, ref, rtol=2e-5, atol=2e-7), (got - ref).abs().max()


def test_both_backward_implementations_agree():
    """The product's omfs_composite_bwd (cross-lane reduction with DPP adds) against the independently written second
    implementations of the same contract in libomfs_experiments.so (the f32 matrix-core reduction; lanes = list entries): on a
    scene with long lists (deep forward, many segments) their 64-byte gradient records agree to 1e-5 of the largest entry per
    column -- float-atomic order noise only."""
    from omfs_4d_video_gen_amd import _lib as L
    from omfs_4d_video_gen_amd.engine import synthetic
    from omfs_4d_video_gen_amd.engine.flame_rig import DeviceFlame, FlameRig
    from omfs_4d_video_gen_amd.engine.gaussians import GaussianModel
    from omfs_4d_video_gen_amd.engine.rasterizer import Rasterizer, make_camera_struct
    N, W, H = 60000, 320, 256
    rig = synthetic.make_rig(4)
    g = synthetic.make_gaussians(N, rig.faces.shape[0], 4)
    seq = synthetic.make_flame_sequence(3, 4)
    dflame = DeviceFlame(FlameRig.from_synthetic(rig), seq)
    ccam = make_camera_struct(synthetic.make_camera(W, H, yaw=0.25), sh_degree=3, bg=(0.1, 0.0, 0.2))
    model, rast = GaussianModel(g), Rasterizer(N, W, H)
    rast.forward(model, dflame.face_frames(1, 1)[1][0], ccam)
    rast._ensure_bwd()
    torch.cuda.synchronize()
    rast.check_status()
    assert int(np.diff(rast.tile_start.cpu().numpy().astype(np.int64)).max()) > 1024       # deep lists present
    rast.dimage.copy_(torch.randn(3, H, W, generator=torch.Generator().manual_seed(3)).cuda())
    gb = L.GradBuffersC(L.ptr(rast.dsplat), 0, L.ptr(rast.dimage), 0, 0, 0)
    out = {}
    for impl in L.BWD_IMPLS:
        rast.dsplat.zero_()
        L.composite_bwd(impl, ccam, rast.rb, gb, L.stream_ptr())
        torch.cuda.synchronize()
        out[impl] = rast.dsplat.cpu().numpy().copy()
    a = out["dpp"][:, :9]
    assert np.abs(a).max() > 0
    for impl in L.BWD_IMPLS[1:]:
        b = out[impl][:, :9]
        for q in range(9):
            scale = np.abs(a[:, q]).max()
            assert np.abs(a[:, q] - b[:, q]).max() <= 1e-5 * scale + 1e-12, (impl, q, np.abs(a[:, q] - b[:, q]).max(), scale)


def test_quadrant_depth_table_is_exact_and_its_hint_changes_nothing():
    """omfs_raster_buffers.quad_depth (ABI 7): the forward leaves the deepest last contributor of every (tile, quadrant) -- equal to
    the maximum of n_contrib over the quadrant's pixels -- and, when the caller keeps the table from one visit of a view to the
    next, reads it first as a priority hint.  Whatever the table held before (zeros, the true depths, garbage), image, final_T,
    n_contrib and the backward's gradient records are the same bits (the records up to the order of their float atomics)."""
    from omfs_4d_video_gen_amd import _lib as L
    from omfs_4d_video_gen_amd.engine import synthetic
    from omfs_4d_video_gen_amd.engine.flame_rig import DeviceFlame, FlameRig
    from omfs_4d_video_gen_amd.engine.gaussians import GaussianModel
    from omfs_4d_video_gen_amd.engine.rasterizer import Rasterizer, make_camera_struct
    N, W, H = 60000, 330, 250          # not a multiple of the tile size: quadrants outside the image exist
    rig = synthetic.make_rig(6)
    g = synthetic.make_gaussians(N, rig.faces.shape[0], 6)
    dflame = DeviceFlame(FlameRig.from_synthetic(rig), synthetic.make_flame_sequence(3, 6))
    ccam = make_camera_struct(synthetic.make_camera(W, H, yaw=-0.3), sh_degree=3, bg=(0.0, 0.1, 0.0))
    model, rast = GaussianModel(g), Rasterizer(N, W, H)
    fxf = dflame.face_frames(2, 1)[1][0]
    rast._ensure_bwd()
    rast.dimage.copy_(torch.randn(3, H, W, generator=torch.Generator().manual_seed(9)).cuda())
    gb = L.GradBuffersC(L.ptr(rast.dsplat), 0, L.ptr(rast.dimage), 0, 0, 0)
    qd = torch.zeros(rast.n_tiles, 4, dtype=torch.int32, device="cuda")
    ref = None
    for fill in ("library-owned", "zeros", "kept", "garbage"):
        if fill == "library-owned":
            rast.rb.quad_depth = 0
        else:
            rast.rb.quad_depth = L.ptr(qd)
            if fill == "garbage":
                qd.copy_(torch.randint(0, 1 << 30, qd.shape, dtype=torch.int32, generator=torch.Generator().manual_seed(1)).cuda())
        rast.forward(model, fxf, ccam)
        rast.dsplat.zero_()
        L.check(L.load().omfs_composite_bwd(ccam, rast.rb, gb, L.stream_ptr()), "omfs_composite_bwd")
        torch.cuda.synchronize()
        rast.check_status()
        got = (rast.image.cpu().numpy().copy(), rast.final_T.cpu().numpy().copy(), rast.n_contrib.cpu().numpy().copy(), rast.dsplat.cpu().numpy().copy())
        if fill != "library-owned":
            # the table now holds the exact depths: per (tile, quadrant) the maximum of n_contrib over its pixels
            nc = np.zeros((rast.gy * 16, rast.gx * 16), np.int64)
            nc[:H, :W] = got[2].view(np.uint32)
            want = nc.reshape(rast.gy, 2, 8, rast.gx, 2, 8).max(axis=(2, 5)).transpose(0, 2, 1, 3).reshape(rast.n_tiles, 4)   # [tile][qy*2+qx]
            assert np.array_equal(qd.cpu().numpy().view(np.uint32).astype(np.int64), want), fill
        if ref is None:
            ref = got
            continue
        for a, b in zip(ref[:3], got[:3]):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), fill
        scale = np.abs(ref[3]).max(0) + 1e-30
        assert (np.abs(ref[3] - got[3]).max(0) <= 1e-5 * scale).all(), fill
    rast.rb.quad_depth = 0


def test_backward_without_its_work_tables():
    """The backward's work tables live in `keys` when the caller passes no quad_depth buffer, and only when they fit: a pair
    capacity that is tiny against the tile count (few Gaussians on a 1080p image) leaves room for the segment table but not for
    the quadrant depths, or for neither -- then composite_bwd falls back to the exit test on the pixels' n_contrib and to the
    bisection of order_seg0.  All three set-ups (both tables / segment table only / none) give the same gradient records."""
    from omfs_4d_video_gen_amd import _lib as L
    from omfs_4d_video_gen_amd.engine import synthetic
    from omfs_4d_video_gen_amd.engine.flame_rig import DeviceFlame, FlameRig
    from omfs_4d_video_gen_amd.engine.gaussians import GaussianModel
    from omfs_4d_video_gen_amd.engine.rasterizer import Rasterizer, make_camera_struct
    N, W, H = 120, 1920, 1080
    rig = synthetic.make_rig(2)
    g = synthetic.make_gaussians(N, rig.faces.shape[0], 2)
    g["log_scale"] = g["log_scale"] + 1.0          # a few large splats: lists of several entries, few pairs in all
    dflame = DeviceFlame(FlameRig.from_synthetic(rig), synthetic.make_flame_sequence(2, 2))
    ccam = make_camera_struct(synthetic.make_camera(W, H, yaw=0.1), sh_degree=3, bg=(0.2, 0.2, 0.2))
    model = GaussianModel(g)
    fxf = dflame.face_frames(1, 1)[1][0]
    dimage = torch.randn(3, H, W, generator=torch.Generator().manual_seed(4)).cuda()
    big = Rasterizer(N, W, H)
    big.forward(model, fxf, ccam)
    torch.cuda.synchronize()
    D, n_tiles = int(big.tile_start[-1]), big.n_tiles
    assert 0 < D < 3800, D
    outs = {}
    for name, cap in (("both tables", None), ("segment table only", 12000), ("no table", 4000)):
        r = Rasterizer(N, W, H, dup_capacity=cap)
        seg_cap = r.seg_capacity
        fits_seg, fits_depth = 2 * r.dup_capacity >= seg_cap, 2 * r.dup_capacity >= seg_cap + 4 * n_tiles
        assert (fits_seg, fits_depth) == {"both tables": (True, True), "segment table only": (True, False), "no table": (False, False)}[name]
        r.forward(model, fxf, ccam)
        r._ensure_bwd()
        r.dimage.copy_(dimage)
        r.dsplat.zero_()
        gb = L.GradBuffersC(L.ptr(r.dsplat), 0, L.ptr(r.dimage), 0, 0, 0)
        for impl in L.BWD_IMPLS:
            r.dsplat.zero_()
            L.composite_bwd(impl, ccam, r.rb, gb, L.stream_ptr())
            torch.cuda.synchronize()
            r.check_status()
            outs[(name, impl)] = r.dsplat.cpu().numpy()[:, :9].copy()
    ref = outs[("both tables", "dpp")]
    assert np.abs(ref).max() > 0
    scale = np.abs(ref).max(0) + 1e-30
    for key, got in outs.items():
        assert (np.abs(got - ref).max(0) <= 1e-5 * scale).all(), key


def test_in_place_sh_adam_is_bit_identical_to_the_plain_step():
    """One GPU: omfs_project_bwd with gb->drgb_out + gb->dir_out leaves the 45 SH gradient planes of degree >= 1 out, and
    omfs_adam_step_sh_rest forms Y_k(dir) * drgb where it updates them.  From the same gradient records (project_bwd has no
    atomics: it is a function of its inputs) both routes -- all 59 planes through the gradient buffer and one Adam launch, or
    14 planes + the in-place step -- give the SAME BITS in parameters and both moments, for every SH degree."""
    from omfs_4d_video_gen_amd import _lib as L
    from omfs_4d_video_gen_amd.engine import synthetic
    from omfs_4d_video_gen_amd.engine.flame_rig import DeviceFlame, FlameRig
    from omfs_4d_video_gen_amd.engine.gaussians import GaussianModel, P_SH
    from omfs_4d_video_gen_amd.engine.rasterizer import Adam, Rasterizer, default_lr_planes, make_camera_struct
    import ctypes as C
    N, W, H = 20000, 200, 150
    rig = synthetic.make_rig(8)
    g = synthetic.make_gaussians(N, rig.faces.shape[0], 8)
    dflame = DeviceFlame(FlameRig.from_synthetic(rig), synthetic.make_flame_sequence(2, 8))
    fxf = dflame.face_frames(1, 1)[1][0]
    lib, s = L.load(), L.stream_ptr()
    for deg in (3, 1, 0):
        ccam = make_camera_struct(synthetic.make_camera(W, H, yaw=0.2), sh_degree=deg, bg=(0.0, 0.0, 0.0))
        model, rast = GaussianModel(g), Rasterizer(N, W, H)
        rast.forward(model, fxf, ccam)
        rast._ensure_bwd()
        rast.dimage.copy_(torch.randn(3, H, W, generator=torch.Generator().manual_seed(deg)).cuda())
        rast.dsplat.zero_()
        gb0 = L.GradBuffersC(L.ptr(rast.dsplat), 0, L.ptr(rast.dimage), 0, 0, 0)
        L.check(lib.omfs_composite_bwd(ccam, rast.rb, gb0, s), "omfs_composite_bwd")
        records = rast.dsplat.clone()                                     # project_bwd consumes (zeroes) the records: both routes start from a copy
        n_pad = model.n_pad
        p0 = model.params.clone()
        gen = torch.Generator().manual_seed(5)
        m0 = (torch.randn(59, n_pad, generator=gen) * 1e-3).cuda()
        v0 = (torch.rand(59, n_pad, generator=gen) * 1e-6).cuda()
        rp = L.RegParamsC(0.01, 1.0, 1.0, 0.6, L.ptr(rast.n_visible))
        gm = rast._gauss(model)
        out = {}
        for route in ("plain", "in place"):
            model.params.copy_(p0)
            opt = Adam(model, default_lr_planes(position_lr=5e-3))
            opt.m.copy_(m0); opt.v.copy_(v0)
            opt.step_count = 6
            grads = torch.zeros(59, n_pad, device="cuda")
            grads[:, :N] = 7.0                                             # stale values in the planes a route does not write
            rast.dsplat.copy_(records)
            drgb, vdir = torch.zeros(3, n_pad, device="cuda"), torch.zeros(3, n_pad, device="cuda")
            gb = L.GradBuffersC(L.ptr(rast.dsplat), L.ptr(grads), L.ptr(rast.dimage), 0, 0,
                                L.ptr(drgb) if route == "in place" else 0, L.ptr(vdir) if route == "in place" else 0)
            L.check(lib.omfs_project_bwd(gm, L.ptr(fxf), ccam, rast.rb, gb, rp, s), "omfs_project_bwd")
            if route == "plain":
                opt.step(grads, 0.5)
            else:
                assert float(grads[P_SH + 3:, :N].min()) == 7.0 == float(grads[P_SH + 3:, :N].max())      # never written
                opt.begin_step(0.5)
                if deg == 1:                                   # the two-launch form
                    opt.apply_planes(grads, 0, P_SH + 3)
                    opt.apply_sh_rest(drgb, vdir, deg)
                else:                                          # one launch, what the trainer issues
                    opt.apply_sh_rest(drgb, vdir, deg, grads_low=grads)
            torch.cuda.synchronize()
            out[route] = (model.params.clone(), opt.m.clone(), opt.v.clone())
        for a, b, name in zip(out["plain"], out["in place"], ("params", "m", "v")):
            assert torch.equal(a.view(torch.int32), b.view(torch.int32)), (deg, name, float((a - b).abs().max()))
        assert not torch.equal(out["plain"][0][P_SH + 3:, :N], p0[P_SH + 3:, :N])          # the step did move the SH planes

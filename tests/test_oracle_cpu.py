"""CPU: the two independent oracle restatements (C bit-level forward, PyTorch differentiable) agree,
and the frozen exp used for the scale activation is accurate.  No GPU, no HIP library calls."""
import numpy as np
import torch

import helpers as H
from omfs_4d_video_gen_amd.engine import synthetic
from omfs_4d_video_gen_amd.engine.flame_rig import FlameRig, DeviceFlame
from omfs_4d_video_gen_amd.engine.gaussians import pack_params
from omfs_4d_video_gen_amd.engine.rasterizer import make_camera_struct
from oracle import c_oracle as CO
from oracle import torch_splat as O


def test_exp_exact_accuracy():
    x = np.linspace(-20, 20, 4001).astype(np.float32)
    got = CO.exp_exact(x).astype(np.float64)
    ref = np.exp(x.astype(np.float64))
    assert np.max(np.abs(got - ref) / ref) < 3e-7


def test_c_oracle_matches_torch_oracle(rig_small):
    rig = rig_small
    n, W, Hh = 1200, 112, 80
    g = synthetic.make_gaussians(n, rig.faces.shape[0], 3)
    seq = synthetic.make_flame_sequence(3, 3)
    cam = synthetic.make_camera(W, Hh, yaw=-0.3)
    dflame = DeviceFlame(FlameRig.from_synthetic(rig), seq, device="cpu")
    ccam = make_camera_struct(cam, sh_degree=3, bg=(0.2, 0.4, 0.6))
    params = pack_params(g)
    c = CO.render(dflame, 2, params, g["binding"], n, CO.camera(ccam))
    t = O.render(H.oracle_rig(rig), H.oracle_gaussians(g), H.oracle_frame(seq, 2), cam, bg=(0.2, 0.4, 0.6), sh_degree=3)
    assert np.allclose(c["verts"], t["verts"].numpy(), rtol=1e-5, atol=2e-7)
    vis = t["proj"]["visible"].numpy()
    assert np.array_equal(c["proj"]["radius"] > 0, vis)
    assert np.allclose(c["proj"]["mean2d"][vis], t["proj"]["mean2d"].numpy()[vis], atol=2e-3)
    assert np.allclose(c["proj"]["rgb"][vis], t["proj"]["rgb"].numpy()[vis], atol=2e-5)
    n_rect = int((c["proj"]["rect"][vis] != t["proj"]["rect"].numpy()[vis]).any(-1).sum())
    assert n_rect <= 1
    l1 = np.abs(c["image"] - t["image"].numpy()).mean()
    assert l1 < 1e-4, l1
    if n_rect == 0:
        lists = [c["ids"][c["tile_start"][i]:c["tile_start"][i + 1]].tolist() for i in range(len(t["lists"]))]
        assert lists == t["lists"]
        assert (c["n_contrib"].astype(np.int64) != t["n_contrib"].numpy()).mean() < 1e-3


def test_oracle_identity_pose_plumbing(rig_small):
    """BASELINE config 1 in miniature: identity FLAME pose, forward splat only, on the CPU."""
    rig = rig_small
    g = synthetic.make_gaussians(500, rig.faces.shape[0], 0)
    seq = synthetic.make_flame_sequence(1, 0, identity=True)
    cam = synthetic.make_camera(64, 64)
    out = O.render(H.oracle_rig(rig), H.oracle_gaussians(g), H.oracle_frame(seq, 0), cam)
    assert torch.allclose(out["verts"], torch.from_numpy(rig.v_template), atol=1e-7)
    assert out["image"].shape == (3, 64, 64) and float(out["image"].max()) > 0.05


def test_config1_at_its_stated_size_on_the_cpu(rig_small):
    """BASELINE config 1 as stated: a single 256x256 frame, 5 k Gaussians, identity FLAME pose, forward splat on the CPU --
    the PyTorch oracle and the C oracle (the bit-level spec the HIP path is held to in tests/test_gpu_bitexact.py) agree."""
    rig = rig_small
    n, W, Hh = 5000, 256, 256
    g = synthetic.make_gaussians(n, rig.faces.shape[0], 0)
    seq = synthetic.make_flame_sequence(1, 0, identity=True)
    cam = synthetic.make_camera(W, Hh, yaw=0.0)
    t = O.render(H.oracle_rig(rig), H.oracle_gaussians(g), H.oracle_frame(seq, 0), cam, sh_degree=3)
    assert torch.allclose(t["verts"], torch.from_numpy(rig.v_template), atol=1e-7)
    dflame = DeviceFlame(FlameRig.from_synthetic(rig), seq, device="cpu")
    ccam = make_camera_struct(cam, sh_degree=3, bg=(0.0, 0.0, 0.0))
    c = CO.render(dflame, 0, pack_params(g), g["binding"], n, CO.camera(ccam))
    assert np.allclose(c["verts"], rig.v_template, atol=1e-7)      # skinning weights sum to 1 up to an ulp
    assert np.abs(c["image"] - t["image"].numpy()).mean() < 1e-4
    assert float(t["image"].max()) > 0.05 and len(c["ids"]) > n


def test_rare_projection_branches_agree_between_the_two_restatements(rig_small):
    """The C oracle and the HIP kernels are operation-for-operation twins, so the independence of the bit-exact check rests on
    the differently formulated PyTorch oracle -- whose robust comparison tolerates 0.5 % of outliers, enough to hide a branch
    only few Gaussians take.  A close-up camera makes those branches common: the head overfills the image (the Jacobian's
    1.3 tan(fov/2) clamp is active for a large share of the Gaussians) and reaches in front of the near plane (t_z <= 0.2 culls);
    on exactly those subsets the two restatements must agree."""
    rig = rig_small
    n, W, Hh = 8000, 160, 128
    g = synthetic.make_gaussians(n, rig.faces.shape[0], 21)
    g["log_scale"] += 1.3                                         # large splats: many centres beyond the clamp still reach the image
    seq = synthetic.make_flame_sequence(2, 21)
    cam = synthetic.make_camera(W, Hh, yaw=0.4, fill=3.5, distance=0.27)
    dflame = DeviceFlame(FlameRig.from_synthetic(rig), seq, device="cpu")
    ccam = make_camera_struct(cam, sh_degree=3, bg=(0.0, 0.0, 0.0))
    c = CO.render(dflame, 1, pack_params(g), g["binding"], n, CO.camera(ccam))
    t = O.render(H.oracle_rig(rig), H.oracle_gaussians(g), H.oracle_frame(seq, 1), cam, sh_degree=3)
    mu = t["proj"]["mean3d"].numpy()
    w2v = np.asarray(cam["world_to_view"], np.float64)
    tv = mu @ w2v[:3, :3].T + w2v[:3, 3]
    near = tv[:, 2] <= 0.2
    safe = np.abs(tv[:, 2] - 0.2) > 1e-4                          # the cull decision itself is not a rounding question there
    clamped = (~near) & ((np.abs(tv[:, 0] / tv[:, 2]) > 1.3 * cam["tanfovx"]) | (np.abs(tv[:, 1] / tv[:, 2]) > 1.3 * cam["tanfovy"]))
    assert near.sum() > n // 50 and clamped.sum() > n // 10, (int(near.sum()), int(clamped.sum()))
    vis_c, vis_t = c["proj"]["radius"] > 0, t["proj"]["visible"].numpy()
    assert not vis_c[near & safe].any() and not vis_t[near & safe].any()           # culled on both sides
    assert np.array_equal(vis_c[safe], vis_t[safe])
    m = clamped & vis_c & vis_t
    assert m.sum() > n // 40, int(m.sum())
    conic_c, conic_t = c["proj"]["conic"][m], t["proj"]["conic"].numpy()[m]
    close = np.abs(conic_c - conic_t) <= 5e-3 * np.abs(conic_t) + 1e-6
    assert close.all(axis=1).mean() > 0.99, float(close.all(axis=1).mean())
    assert np.abs(c["proj"]["radius"][m] - t["proj"]["radius"].numpy()[m]).max() <= 1
    assert np.abs(c["image"] - t["image"].numpy()).mean() < 1e-4


def test_tile_culling_does_not_change_the_image(rig_small):
    """The engine drops (Gaussian, tile) pairs that cannot reach alpha >= 1/255 inside the tile.
    The composite of the culled lists must be BIT-IDENTICAL to the composite of the upstream-style
    lists (every tile of the 3-sigma rectangle): image and final T."""
    rig = rig_small
    n, W, Hh = 6000, 208, 160
    g = synthetic.make_gaussians(n, rig.faces.shape[0], 9)
    g["opacity"][: n // 3] -= 3.0      # many faint splats: the opacity-aware part of the test matters
    seq = synthetic.make_flame_sequence(2, 9)
    dflame = DeviceFlame(FlameRig.from_synthetic(rig), seq, device="cpu")
    for yaw in (0.0, 0.9):
        ccam = make_camera_struct(synthetic.make_camera(W, Hh, yaw=yaw), sh_degree=3, bg=(0.3, 0.6, 0.9))
        a = CO.render(dflame, 1, pack_params(g), g["binding"], n, CO.camera(ccam), cull=True)
        b = CO.render(dflame, 1, pack_params(g), g["binding"], n, CO.camera(ccam), cull=False)
        assert len(a["ids"]) < 0.9 * len(b["ids"]), (len(a["ids"]), len(b["ids"]))
        assert np.array_equal(a["image"].view(np.uint32), b["image"].view(np.uint32))
        assert np.array_equal(a["final_T"].view(np.uint32), b["final_T"].view(np.uint32))


def test_densify_reference_generator_and_layout():
    """oracle/densify_ref.py: the counter-based generator is deterministic, keyed, and standard normal; compact() lays the
    output out as [kept | clones | first children | second children] with fresh moments for everything new."""
    from oracle import densify_ref as R
    assert int(R.mix32(np.uint32(0))) == 0 and len({int(x) for x in R.mix32(np.arange(1, 1000, dtype=np.uint32))}) == 999   # a bijection
    ids = np.arange(200000)
    a, b = R.normal_samples(5, 9, ids, 0), R.normal_samples(5, 9, ids, 0)
    assert np.array_equal(a, b)
    assert not np.array_equal(a, R.normal_samples(5, 9, ids, 1)) and not np.array_equal(a, R.normal_samples(6, 9, ids, 0))
    assert abs(float(a.mean())) < 0.01 and abs(float(a.std()) - 1.0) < 0.01 and np.isfinite(a).all()
    rng = np.random.default_rng(0)
    n = 50
    params = rng.normal(size=(59, n)).astype(np.float32)
    binding = rng.integers(0, 7, n).astype(np.int32)
    m, v = rng.normal(size=(59, n)).astype(np.float32), rng.uniform(size=(59, n)).astype(np.float32)
    cls = np.array([1, 3, 4, 0, 2] * 10, np.uint8)
    p2, b2, m2, v2 = R.compact(params, binding, m, v, cls, 1, 2)
    keep, clone, split = (cls & 1) > 0, (cls & 2) > 0, (cls & 4) > 0
    nk, nc, ns = int(keep.sum()), int(clone.sum()), int(split.sum())
    assert p2.shape == (59, nk + nc + 2 * ns) and b2.shape == (nk + nc + 2 * ns,)
    assert np.array_equal(p2[:, :nk], params[:, keep]) and np.array_equal(p2[:, nk:nk + nc], params[:, clone])
    assert np.array_equal(m2[:, :nk], m[:, keep]) and not m2[:, nk:].any() and not v2[:, nk:].any()
    assert np.array_equal(b2[nk + nc:nk + nc + ns], binding[split]) and np.array_equal(b2[nk + nc + ns:], binding[split])
    assert np.allclose(p2[3:6, nk + nc:nk + nc + ns], params[3:6, split] - np.log(1.6), atol=1e-6)

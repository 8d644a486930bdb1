"""GPU: adaptive density control keeps the engine consistent -- N changes, children inherit the parent's
triangle, buffers are re-padded, training and rendering go on, decisions are reproducible."""
import numpy as np
import pytest
import torch

from omfs_4d_video_gen_amd.engine import synthetic

pytestmark = pytest.mark.gpu


def _trainer(seed=0):
    from omfs_4d_video_gen_amd.engine.flame_rig import FlameRig
    from omfs_4d_video_gen_amd.engine.trainer import Trainer, View
    rig = synthetic.make_rig(0)
    seq = synthetic.make_flame_sequence(4, 0)
    g = synthetic.make_gaussians(6000, rig.faces.shape[0], 0)
    gen = torch.Generator().manual_seed(1)
    views = [View(synthetic.make_camera(128, 96, yaw=0.3 * i - 0.4), i, target=torch.rand(3, 96, 128, generator=gen).cuda()) for i in range(4)]
    return Trainer(FlameRig.from_synthetic(rig), seq, g, views, 128, 96, start_sh_degree=3, n_capacity=30000), rig


def test_densify_prune_reset_and_keep_training():
    from omfs_4d_video_gen_amd.engine.densify import DensityController, scene_extent
    tr, rig = _trainer()
    ctl = DensityController(tr, scene_extent(tr.views), from_iter=5, until_iter=1000, interval=10, grad_threshold=1e-6,
                            opacity_reset_interval=25, max_gaussians=30000, seed=3)
    tr.model.params[10, :500] = -9.0            # 500 nearly transparent Gaussians: must be pruned
    sizes = []
    for it in range(1, 41):
        tr.step()
        ctl.after_step(it)
        sizes.append(tr.model.n)
    torch.cuda.synchronize()
    tr.rast.check_status()
    assert [e["iteration"] for e in ctl.log] == [10, 20, 30, 40]
    first = ctl.log[0]
    assert first["pruned"] >= 500 and first["cloned"] + first["split"] > 0
    assert first["after"] == first["before"] - first["pruned"] - first["split"] + first["cloned"] + 2 * first["split"]
    assert tr.model.n == ctl.log[-1]["after"] and tr.model.n_pad % 256 == 0 and tr.model.n <= 30000
    assert tr.model.params.shape == (59, tr.model.n_pad) == tr.opt.m.shape == tr.grads.shape
    assert torch.isfinite(tr.model.params).all()
    b = tr.model.binding.cpu().numpy()
    assert b.shape == (tr.model.n,) and b.min() >= 0 and b.max() < rig.faces.shape[0]
    assert float(tr.model.params[10, :tr.model.n].max()) < 0.5   # opacity reset at iteration 25 (logit(0.01) = -4.6, then 15 steps)
    assert float(tr.model.params[0:3, tr.model.n:].abs().max()) == 0.0
    img = tr.rast.image.cpu()
    assert torch.isfinite(img).all()


def test_densification_is_deterministic_across_replicas():
    """Two independent trainers (what two ranks hold after the all-reduce) take identical decisions."""
    from omfs_4d_video_gen_amd.engine.densify import DensityController, scene_extent
    outs = []
    for _ in range(2):
        tr, _ = _trainer()
        ctl = DensityController(tr, scene_extent(tr.views), from_iter=0, until_iter=100, interval=5, grad_threshold=1e-6, seed=7)
        for it in range(1, 6):
            tr.step()
        # identical statistics on both replicas (as after an all-reduce)
        tr.densify_stats.copy_(torch.linspace(0, 1e-3, 2 * tr.model.n_pad, device="cuda").reshape(2, -1))
        tr.densify_stats[1].fill_(1.0)
        ctl.densify_and_prune(5)
        outs.append((tr.model.n, tr.model.binding.cpu().numpy().copy(), tr.model.params[3:10].cpu().numpy().copy()))
    assert outs[0][0] == outs[1][0] and np.array_equal(outs[0][1], outs[1][1])
    # scales / rotations of the children depend only on the parents' (already differing by atomics noise) values
    assert np.allclose(outs[0][2], outs[1][2], rtol=1e-3, atol=1e-5)


@pytest.mark.parametrize("size_prune", [False, True])
def test_classify_scan_compact_match_the_numpy_restatement(size_prune):
    """The three HIP entries against oracle/densify_ref.py on the same seeded inputs: classification bits (away from the
    thresholds), output order, parent triangles and Adam moments bit-exact; the split samples to 1e-5."""
    from omfs_4d_video_gen_amd import _lib as L
    from omfs_4d_video_gen_amd.engine.gaussians import GaussianModel
    from oracle import densify_ref as R
    rng = np.random.default_rng(9)
    rig = synthetic.make_rig(0)
    F = rig.faces.shape[0]
    n = 7003
    model = GaussianModel(synthetic.make_gaussians(n, F, 3))
    n_pad = model.n_pad
    face_xf = torch.zeros(F, 16)
    face_xf[:, 12] = torch.from_numpy(rng.uniform(0.002, 0.02, F).astype(np.float32))
    face_xf = face_xf.cuda()
    stats = np.zeros((2, n_pad), np.float32)
    stats[0, :n] = rng.uniform(0, 8e-4, n) * rng.integers(0, 5, n)
    stats[1, :n] = rng.integers(0, 5, n)
    stats_d = torch.from_numpy(stats).cuda()
    adam_m = torch.from_numpy(rng.normal(size=(59, n_pad)).astype(np.float32)).cuda()
    adam_v = torch.from_numpy(rng.uniform(size=(59, n_pad)).astype(np.float32)).cuda()
    model.params[10, :n] = torch.from_numpy(rng.uniform(-7, 4, n).astype(np.float32)).cuda()     # some nearly transparent
    size_thr, prune_size = 0.004, (0.012 if size_prune else 0.0)
    dp = L.DensifyParamsC(2e-4, size_thr, 0.005, prune_size, 1234, 77)
    lib, s = L.load(), L.stream_ptr()
    g = L.GaussiansC(n, n_pad, L.ptr(model.params), L.ptr(model.binding))
    nb = (n + 255) // 256
    cls = torch.empty(n, dtype=torch.uint8, device="cuda")
    grads = torch.empty(n, device="cuda")
    counts = torch.empty(3, nb, dtype=torch.int32, device="cuda")
    totals = torch.empty(3, dtype=torch.int32, device="cuda")
    L.check(lib.omfs_densify_classify(g, L.ptr(face_xf), L.ptr(stats_d), dp, L.ptr(cls), L.ptr(grads), L.ptr(counts), s), "classify")
    L.check(lib.omfs_densify_scan(L.ptr(counts), n, L.ptr(totals), s), "scan")
    p_h, b_h = model.params.cpu().numpy()[:, :n], model.binding.cpu().numpy()
    want_cls, margin = R.classify(p_h, b_h, face_xf[:, 12].cpu().numpy(), stats[:, :n], 2e-4, size_thr, 0.005, prune_size)
    got_cls = cls.cpu().numpy()
    clear = margin > 1e-5
    assert clear.mean() > 0.99 and np.array_equal(got_cls[clear], want_cls[clear])
    assert np.array_equal(grads.cpu().numpy(), stats[0, :n] / np.maximum(stats[1, :n], 1.0))
    tk, tc, ts = [int(x) for x in totals.tolist()]
    assert (tk, tc, ts) == (int((got_cls & 1).astype(bool).sum()), int((got_cls & 2).astype(bool).sum()), int((got_cls & 4).astype(bool).sum()))
    assert min(tk, tc, ts) > 100 and tk < n                                  # every class is exercised
    n_out = tk + tc + 2 * ts
    n_out_pad = (n_out + 255) // 256 * 256
    new_p = torch.zeros(59, n_out_pad, device="cuda")
    new_m, new_v = torch.zeros_like(new_p), torch.zeros_like(new_p)
    new_b = torch.full((n_out,), -1, dtype=torch.int32, device="cuda")
    L.check(lib.omfs_densify_compact(g, L.ptr(adam_m), L.ptr(adam_v), L.ptr(cls), L.ptr(counts), L.ptr(totals), dp, n_out_pad,
                                     L.ptr(new_p), L.ptr(new_b), L.ptr(new_m), L.ptr(new_v), s), "compact")
    torch.cuda.synchronize()
    wp, wb, wm, wv = R.compact(p_h, b_h, adam_m.cpu().numpy()[:, :n], adam_v.cpu().numpy()[:, :n], got_cls, 1234, 77)
    assert wp.shape[1] == n_out
    gp = new_p.cpu().numpy()
    assert np.array_equal(new_b.cpu().numpy(), wb)
    assert np.array_equal(new_m.cpu().numpy()[:, :n_out], wm) and np.array_equal(new_v.cpu().numpy()[:, :n_out], wv)
    assert np.array_equal(gp[:, :tk + tc], wp[:, :tk + tc])                     # kept and cloned: copies
    assert not gp[:, n_out:].any()
    ch_g, ch_w = gp[:, tk + tc:n_out], wp[:, tk + tc:]
    assert np.array_equal(ch_g[6:], ch_w[6:])                                    # rotation, opacity, SH: the parent's
    assert np.allclose(ch_g[3:6], ch_w[3:6], atol=1e-6)                          # log-scale - log 1.6
    assert np.allclose(ch_g[0:3], ch_w[0:3], atol=1e-5, rtol=1e-5)              # sampled positions
    # the two children of one parent differ, and the samples have the parent's spread
    d = (ch_g[0:3, :ts] - ch_g[0:3, ts:])
    assert (np.abs(d).sum(0) > 0).all()

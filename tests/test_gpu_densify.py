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

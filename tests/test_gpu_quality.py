"""GPU quality bar in the shape of the reference's own regression (`02_Visual_Engine/single_frame_experiment.py:84-173`:
train 3000 iterations on one frame, render it, compare with the ground truth): a 20 000-Gaussian head rendered by the
engine is the ground truth, a faint grey mesh-bound cloud is trained on it for 3000 iterations, and the render must
reach PSNR >= 30 dB -- once with the Gaussian count fixed, once with adaptive density control switched on."""
import math

import numpy as np
import pytest
import torch

from omfs_4d_video_gen_amd.engine import synthetic

pytestmark = pytest.mark.gpu

N, W, H, ITERS = 20000, 320, 256, 3000


def _ground_truth(rig_faces: int, face_centres: np.ndarray) -> dict:
    """A head with structure a camera would see: colour varies smoothly over the surface (plus a little per-Gaussian
    variation), the splats are opaque and roughly triangle-sized."""
    g = synthetic.make_gaussians(N, rig_faces, 5)
    c = face_centres[g["binding"]]
    base = 0.5 + 0.35 * np.stack([np.sin(40.0 * c[:, 0] + 1.0), np.sin(33.0 * c[:, 1]), np.sin(47.0 * c[:, 2] + 2.0)], 1)
    rng = np.random.default_rng(9)
    g["sh"][:, 0, :] = ((base + 0.03 * rng.standard_normal((N, 3)) - 0.5) / 0.28209479177387814).astype(np.float32)
    g["sh"][:, 1:, :] *= 0.3
    g["opacity"][:] = rng.uniform(1.0, 4.0, N).astype(np.float32)
    g["log_scale"][:] = rng.uniform(math.log(0.3), math.log(0.7), (N, 3)).astype(np.float32)
    return g


@pytest.mark.parametrize("densify", [False, True])
def test_three_thousand_iterations_on_one_frame_reach_30_db(densify):
    from omfs_4d_video_gen_amd.engine.densify import DensityController
    from omfs_4d_video_gen_amd.engine.flame_rig import FlameRig
    from omfs_4d_video_gen_amd.engine.train import initial_gaussians
    from omfs_4d_video_gen_amd.engine.trainer import Renderer, Trainer, View
    srig = synthetic.make_rig(0)
    rig = FlameRig.from_synthetic(srig)
    seq = synthetic.make_flame_sequence(2, 0)
    cam = synthetic.make_camera(W, H, yaw=0.15, fill=0.8)
    centres = srig.v_template[srig.faces].mean(1)
    gt = _ground_truth(rig.n_faces, centres)
    view = View(cam, 1)
    bg = (1.0, 1.0, 1.0)
    view.target = Renderer(rig, seq, gt, W, H, bg=bg).render(view).clone()
    n0 = N // 2 if densify else N
    tr = Trainer(rig, seq, initial_gaussians(n0, rig.n_faces, 0), [view], W, H, bg=bg, iterations=ITERS, start_sh_degree=0,
                 sh_increase_every=500, n_capacity=2 * N, finetune_flame=False)
    ctl = None
    if densify:
        ctl = DensityController(tr, 1.1, from_iter=300, until_iter=2400, interval=150, grad_threshold=2e-4,
                                opacity_reset_interval=100000, max_gaussians=2 * N, seed=0)
    for it in range(1, ITERS + 1):
        tr.step()
        if ctl is not None:
            ctl.after_step(it)
    torch.cuda.synchronize()
    tr.rast.check_status()
    out = Renderer(rig, seq, tr.model.to_dict(), W, H, bg=bg).render(view)
    mse = float(((out - view.target) ** 2).mean())
    psnr = 10.0 * math.log10(1.0 / max(mse, 1e-12))
    print(f"PSNR after {ITERS} iterations{' with density control' if densify else ''}: {psnr:.2f} dB, {tr.model.n} Gaussians")
    if densify:
        assert tr.model.n > n0 and len(ctl.log) >= 5, ctl.log[-3:]
    assert psnr >= 30.0, psnr

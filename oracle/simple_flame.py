"""ORACLE (test infrastructure): PyTorch-CPU restatement of the reference's SimpleFLAME landmark
model and fit loop, `02_Visual_Engine/flame_fitter.py:122-197` (forward) and `:294-444` (fit).
PINNED: checked against tests/golden/flame_fitter_golden.npz, which was produced by running the
reference itself (tests/golden/make_goldens.py).  Evaluates all V vertices and then mixes three per
landmark, exactly as the reference does (the product folds the mix into the basis instead)."""
from __future__ import annotations

import numpy as np
import torch

from .torch_splat import rodrigues


class SimpleFlameOracle:
    def __init__(self, rig, n_shape=100, n_expr=50):
        """rig: synthetic.SyntheticRig (same arrays the pickle holds)."""
        self.n_shape, self.n_expr = n_shape, n_expr
        f = lambda a: torch.from_numpy(np.asarray(a, np.float32))
        self.v_template = f(rig.v_template)
        self.sd_shape = f(rig.shapedirs[:, :, :n_shape])          # flame_fitter.py:91
        self.sd_expr = f(rig.shapedirs[:, :, 300:300 + n_expr])   # :92
        self.faces = torch.from_numpy(rig.faces.astype(np.int64))
        self.lmk_faces_idx = torch.from_numpy(np.asarray(rig.lmk_faces_idx, np.int64))
        self.lmk_bary = f(rig.lmk_bary_coords)

    def forward(self, shape, expr, rotation, jaw, translation):
        v = self.v_template.unsqueeze(0) + torch.einsum("ijk,bk->bij", self.sd_shape, shape) + torch.einsum("ijk,bk->bij", self.sd_expr, expr)  # :169-175
        lower = (self.v_template[:, 1] < self.v_template[:, 1].mean()).float()          # :179
        off = torch.zeros_like(v)
        off[:, :, 1] = -jaw[:, 0:1] * lower.unsqueeze(0) * 0.15                            # :181
        v = v + off
        v = torch.bmm(v, rodrigues(rotation).transpose(1, 2)) + translation.unsqueeze(1)  # :185-189
        tri = v[:, self.faces[self.lmk_faces_idx]]                                         # :192-193
        return (tri * self.lmk_bary.unsqueeze(0).unsqueeze(-1)).sum(dim=2)                 # :194-195


def fit(oracle: SimpleFlameOracle, lmk2d, valid, image_size, init_rot, lr=0.01, n_iters=3):
    """The fit loop of flame_fitter.py:339-413 on given (T,68,2) pixel landmarks / validity mask."""
    T = lmk2d.shape[0]
    W, H = image_size
    P = torch.nn.Parameter
    shape, expr = P(torch.zeros(1, oracle.n_shape)), P(torch.zeros(T, oracle.n_expr))
    rotation, jaw, translation = P(torch.tensor(init_rot, dtype=torch.float32)), P(torch.zeros(T, 3)), P(torch.zeros(T, 3))
    target = torch.zeros(T, 68, 2)
    with torch.no_grad():
        translation[:, 2] = -5.0                                                           # :347
        for i in range(T):
            if valid[i]:
                translation[i, 0] = float(lmk2d[i, :, 0].mean() / W * 2 - 1) * 2            # :351-354
                translation[i, 1] = float(lmk2d[i, :, 1].mean() / H * 2 - 1) * 2
                target[i, :, 0] = torch.tensor(lmk2d[i, :, 0] / W * 2 - 1)                   # :370-371
                target[i, :, 1] = torch.tensor(lmk2d[i, :, 1] / H * 2 - 1)
    vmask = torch.tensor(np.asarray(valid, bool))
    opt = torch.optim.Adam([{"params": shape, "lr": lr * 0.1}, {"params": expr, "lr": lr}, {"params": rotation, "lr": lr * 0.3},
                            {"params": jaw, "lr": lr}, {"params": translation, "lr": lr * 0.5}])           # :356-362
    n_lmk = min(68, oracle.lmk_faces_idx.shape[0])
    for _ in range(n_iters):
        opt.zero_grad()
        l3 = oracle.forward(shape.expand(T, -1), expr, rotation, jaw, translation)
        px = l3[:, :n_lmk, 0] / (-l3[:, :n_lmk, 2] + 1e-8)                                 # :385-386
        py = l3[:, :n_lmk, 1] / (-l3[:, :n_lmk, 2] + 1e-8)
        diff = (torch.stack([px, py], -1) - target[:, :n_lmk]) ** 2
        loss = (diff * vmask[:, None, None]).sum() / max(int(vmask.sum()) * n_lmk, 1)      # :392
        loss = loss + (shape ** 2).mean() * 0.001 + (expr ** 2).mean() * 0.0001 + (jaw ** 2).mean() * 0.001   # :395-397
        if T > 1:
            for x in (expr, jaw, rotation, translation):
                loss = loss + ((x[1:] - x[:-1]) ** 2).mean() * 0.001                        # :401-404
        loss.backward()
        opt.step()
    return {"shape": shape.detach().numpy()[0], "expr": expr.detach().numpy(), "rotation": rotation.detach().numpy(),
            "jaw_pose": jaw.detach().numpy(), "translation": translation.detach().numpy()}

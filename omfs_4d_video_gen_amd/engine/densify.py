"""Adaptive density control for mesh-bound Gaussians (the absent upstream
`train.py` does this between iterations, call site `02_Visual_Engine/train_ghost.py:227-271`).

Every `interval` iterations between `from_iter` and `until_iter`:
  * mean view-space positional gradient per Gaussian (statistics accumulated by `omfs_project_bwd`);
  * clone  : gradient >= threshold and world size <= percent_dense * extent  -> append a copy;
  * split  : gradient >= threshold and world size >  percent_dense * extent  -> replace by 2 samples drawn
             from the Gaussian itself (local frame), scales divided by 1.6;
  * prune  : opacity < min_opacity, or world size > 0.1 * extent once opacities have been reset;
  * children inherit the parent's triangle (`binding`).
Every `opacity_reset_interval` iterations opacities are clamped to 0.01 and their Adam moments cleared.

This runs a handful of times per training run, so it is written with torch tensor ops on the SoA
(stream compaction = boolean indexing); the per-iteration cost is the two extra words per Gaussian that
`project_bwd` accumulates.  Data parallel: the statistics are all-reduced and the split samples come from
a generator seeded with (seed, iteration), so every rank takes identical decisions and N stays equal.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from .gaussians import NPLANES, P_OPACITY, P_ROT, P_SCALE, P_XYZ


def _pad(n: int) -> int:
    return (n + 255) // 256 * 256


def _quat_to_rotmat(q: torch.Tensor) -> torch.Tensor:
    q = q / q.norm(dim=0, keepdim=True)
    r, x, y, z = q[0], q[1], q[2], q[3]
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1).reshape(-1, 3, 3)


class DensityController:
    def __init__(self, trainer, extent: float, from_iter=10000, until_iter=600000, interval=2000, grad_threshold=2e-4,
                 percent_dense=0.01, min_opacity=0.005, opacity_reset_interval=60000, max_gaussians=None, seed=0):
        self.t = trainer
        self.extent = float(extent)
        self.from_iter, self.until_iter, self.interval = from_iter, until_iter, interval
        self.grad_threshold, self.percent_dense, self.min_opacity = grad_threshold, percent_dense, min_opacity
        self.opacity_reset_interval = opacity_reset_interval
        self.max_gaussians = int(max_gaussians if max_gaussians else trainer.rast.n_capacity)
        self.seed = seed
        self.log = []
        trainer.densify_stats = torch.zeros(2, trainer.model.n_pad, device=trainer.device)

    # ------------------------------------------------------------------ called once per iteration, after step()
    def after_step(self, iteration: int) -> None:
        if iteration < self.until_iter and iteration > self.from_iter and iteration % self.interval == 0:
            self.densify_and_prune(iteration, size_prune=iteration > self.opacity_reset_interval)
        if iteration % self.opacity_reset_interval == 0 and iteration < self.until_iter:
            self.reset_opacity()

    def reset_opacity(self) -> None:
        m = self.t.model
        cap = math.log(0.01 / 0.99)
        m.params[P_OPACITY, :m.n].clamp_(max=cap)
        self.t.opt.m[P_OPACITY].zero_()
        self.t.opt.v[P_OPACITY].zero_()

    def densify_and_prune(self, iteration: int, size_prune: bool = False) -> dict:
        t, m = self.t, self.t.model
        n = m.n
        stats = t.densify_stats
        if t.world > 1:
            from .distributed import allreduce_sum_
            allreduce_sum_(stats, t.pg)
        grads = stats[0, :n] / stats[1, :n].clamp(min=1.0)
        p = m.params[:, :n]
        face_scale = t.dflame.face_frames(0, 1)[1][0, :, 12]                     # world size of each triangle
        world_max = torch.exp(p[P_SCALE:P_SCALE + 3]).amax(0) * face_scale[m.binding.long()]
        hot = grads >= self.grad_threshold
        small = world_max <= self.percent_dense * self.extent
        room = max(0, self.max_gaussians - n)
        clone = hot & small
        split = hot & ~small
        # respect the capacity: keep the strongest gradients
        want = int(clone.sum()) + int(split.sum())
        if want > room:
            order = torch.argsort(torch.where(hot, grads, torch.zeros_like(grads)), descending=True)[:room]
            keep = torch.zeros_like(hot)
            keep[order] = True
            clone &= keep
            split &= keep
        gen = torch.Generator(device="cpu").manual_seed(self.seed * 1_000_003 + iteration)
        ns = int(split.sum())
        sp = p[:, split]
        stds = torch.exp(sp[P_SCALE:P_SCALE + 3])                                # local (triangle-relative) scales
        noise = torch.randn(2, 3, ns, generator=gen).to(p.device) * stds.unsqueeze(0)
        R = _quat_to_rotmat(sp[P_ROT:P_ROT + 4])                                   # (ns,3,3)
        children = []
        for k in range(2):
            c = sp.clone()
            c[P_XYZ:P_XYZ + 3] = sp[P_XYZ:P_XYZ + 3] + torch.einsum("nij,jn->in", R, noise[k])
            c[P_SCALE:P_SCALE + 3] = torch.log(stds / 1.6)
            children.append(c)
        opacity = torch.sigmoid(p[P_OPACITY])
        prune = split | (opacity < self.min_opacity)
        if size_prune:
            prune |= world_max > 0.1 * self.extent
        keep_mask = ~prune
        new_p = torch.cat([p[:, keep_mask], p[:, clone]] + children, 1)
        sb = m.binding[split]
        new_b = torch.cat([m.binding[keep_mask], m.binding[clone], sb, sb])
        zeros = lambda k: torch.zeros(NPLANES, k, device=p.device)
        new_m = torch.cat([t.opt.m[:, :n][:, keep_mask], zeros(int(clone.sum()) + 2 * ns)], 1)
        new_v = torch.cat([t.opt.v[:, :n][:, keep_mask], zeros(int(clone.sum()) + 2 * ns)], 1)
        info = {"iteration": iteration, "before": n, "cloned": int(clone.sum()), "split": ns,
                "pruned": int(prune.sum()) - ns, "after": int(new_p.shape[1])}
        self._install(new_p, new_b, new_m, new_v)
        self.log.append(info)
        return info

    def _install(self, params, binding, adam_m, adam_v) -> None:
        t, m = self.t, self.t.model
        n = int(params.shape[1])
        if n > t.rast.n_capacity:
            raise RuntimeError(f"{n} Gaussians exceed the rasteriser capacity {t.rast.n_capacity}")
        if n == 0:
            raise RuntimeError("densification pruned every Gaussian")
        n_pad = _pad(n)

        def padded(x, fill_rot=False):
            out = torch.zeros(NPLANES, n_pad, device=x.device)
            out[:, :n] = x
            if fill_rot:
                out[P_ROT, n:] = 1.0
            return out
        m.params, m.binding, m.n, m.n_pad = padded(params, True), binding.to(torch.int32).contiguous(), n, n_pad
        t.opt.m, t.opt.v = padded(adam_m), padded(adam_v)
        t.grads = torch.zeros(NPLANES, n_pad, device=params.device)
        t.densify_stats = torch.zeros(2, n_pad, device=params.device)
        # stale projections beyond the new count must not look visible
        t.rast.g2[n:].zero_()


def scene_extent(views: list) -> float:
    """3DGS `cameras_extent`: 1.1 x the largest distance of a camera from the mean camera position."""
    pos = np.stack([np.asarray(v.camera["cam_pos"], np.float64) for v in views])
    return float(np.linalg.norm(pos - pos.mean(0), axis=1).max() * 1.1) or 1.0

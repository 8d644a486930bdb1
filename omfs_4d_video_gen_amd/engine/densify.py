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

The device work is three HIP launches behind the C ABI (`omfs_densify_classify`, `_scan`, `_compact`: classification,
offsets, stream compaction of parameters / parent triangles / Adam moments and the split samples); this module decides
when to run them, sizes the new buffers and installs them.  The per-iteration cost is the two extra words per Gaussian
that `project_bwd` accumulates.  Data parallel: the statistics are all-reduced and the split samples come from a
counter-based generator keyed by (seed, iteration, index), so every rank takes identical decisions and N stays equal.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from .. import _lib as L
from .gaussians import NPLANES, P_OPACITY, P_ROT


def _pad(n: int) -> int:
    return (n + 255) // 256 * 256


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
        n, dev = m.n, m.params.device
        t.sync_optimizer_state()       # "sharded" exchange: the compaction below reads every Gaussian's moments
        stats = t.densify_stats
        if t.world > 1:
            from .distributed import allreduce_sum_
            allreduce_sum_(stats, t.pg)
        lib, s = L.load(), L.stream_ptr()
        face_xf = t.dflame.face_frames(0, 1)[1][0]                    # word 12 of a record: world size of the triangle
        g = L.GaussiansC(n, m.n_pad, L.ptr(m.params), L.ptr(m.binding))
        n_blocks = (n + 255) // 256
        cls = torch.empty(n, dtype=torch.uint8, device=dev)
        grads = torch.empty(n, device=dev)
        counts = torch.empty(3, n_blocks, dtype=torch.int32, device=dev)
        totals = torch.empty(3, dtype=torch.int32, device=dev)
        dp = L.DensifyParamsC(self.grad_threshold, self.percent_dense * self.extent, self.min_opacity,
                              0.1 * self.extent if size_prune else 0.0, int(iteration) & 0xFFFFFFFF, int(self.seed) & 0xFFFFFFFF)

        def classify():
            L.check(lib.omfs_densify_classify(g, L.ptr(face_xf), L.ptr(stats), dp, L.ptr(cls), L.ptr(grads), L.ptr(counts), s),
                    "omfs_densify_classify")
            L.check(lib.omfs_densify_scan(L.ptr(counts), n, L.ptr(totals), s), "omfs_densify_scan")
            return [int(x) for x in totals.tolist()]                  # host sync: the new buffers are sized from it

        n_keep, n_clone, n_split = classify()
        room = max(0, self.max_gaussians - n)
        if n_clone + n_split > room:
            # respect the capacity: only the `room` strongest gradients densify -- the threshold is raised to just above
            # the (room+1)-th largest mean gradient and the classification repeated
            top = torch.topk(grads, room + 1).values
            dp.grad_threshold = float(np.nextafter(np.float32(top[-1].item()), np.float32(np.inf)))
            n_keep, n_clone, n_split = classify()
        n_out = n_keep + n_clone + 2 * n_split
        if n_out > t.rast.n_capacity:
            raise RuntimeError(f"{n_out} Gaussians exceed the rasteriser capacity {t.rast.n_capacity}")
        if n_out == 0:
            raise RuntimeError("densification pruned every Gaussian")
        n_pad = _pad(n_out)
        new_p = torch.zeros(NPLANES, n_pad, device=dev)
        new_m, new_v = torch.zeros_like(new_p), torch.zeros_like(new_p)
        new_b = torch.zeros(n_out, dtype=torch.int32, device=dev)
        L.check(lib.omfs_densify_compact(g, L.ptr(t.opt.m), L.ptr(t.opt.v), L.ptr(cls), L.ptr(counts), L.ptr(totals), dp, n_pad,
                                         L.ptr(new_p), L.ptr(new_b), L.ptr(new_m), L.ptr(new_v), s), "omfs_densify_compact")
        new_p[P_ROT, n_out:] = 1.0                                    # keep padded quaternions normalisable
        info = {"iteration": iteration, "before": n, "cloned": n_clone, "split": n_split,
                "pruned": n - n_keep - n_split, "after": n_out}
        self._install(new_p, new_b, new_m, new_v, n_out)
        self.log.append(info)
        return info

    def _install(self, params, binding, adam_m, adam_v, n: int) -> None:
        """Swap in the compacted buffers ([59][n_pad] / [n]) and everything sized by them."""
        t, m = self.t, self.t.model
        n_pad = int(params.shape[1])
        m.params, m.binding, m.n, m.n_pad = params, binding, n, n_pad
        m.order = None                 # the compaction re-indexed the cloud: the storage order is the order from here on
        t.opt.m, t.opt.v = adam_m, adam_v
        t.alloc_grads(n_pad)
        t.densify_stats = torch.zeros(2, n_pad, device=params.device)
        t.invalidate_graphs()          # captured iterations refer to the buffers just replaced
        # stale projections beyond the new count must not look visible
        t.rast.g2[n:].zero_()


def scene_extent(views: list) -> float:
    """3DGS `cameras_extent`: 1.1 x the largest distance of a camera from the mean camera position."""
    pos = np.stack([np.asarray(v.camera["cam_pos"], np.float64) for v in views])
    return float(np.linalg.norm(pos - pos.mean(0), axis=1).max() * 1.1) or 1.0

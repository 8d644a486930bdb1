"""EXPERIMENTAL: whole training iterations as hipGraphs (`OMFS_STEP_GRAPH=1`, single GPU).

Kept apart from the training loop (`engine/trainer.py`) because it is not the path anything runs by default: on ROCm 7.2 a replayed
iteration is SLOWER than the eager C-ABI calls at every size measured (5 k Gaussians / 256 x 256: 4335 it/s eager, 3662 replayed, 3845
with four iterations per graph).  What it demonstrates -- and `tests/test_gpu_trajectory.py` holds against the
eager trajectory -- is that an iteration contains no host-computed scalar: the step-dependent values (position learning rate, Adam
bias corrections, the next view's FLAME timestep) live in an `omfs_step_state` on the device, advanced by the graph's first node
(`omfs_step_advance`, `omfs_adam_step_dev`, `omfs_adam_flat_multi(state_dev)`), so a host may capture a step of its own.
"""
from __future__ import annotations

import os

import torch

from .. import _lib as L
from .gaussians import NPLANES


class GraphReplayMixin:
    """`_graph_key`, `_graph_eligible`, `_capture_step`, `_step_graph` of `Trainer` (which owns the state they use)."""

    def _graph_key(self, it: int):
        from .distributed import view_index
        vi = view_index(it, self.rank, self.world, len(self.views), self.view_seed)
        return (vi, it & 1, self.sh_degree, self.model.n, self.model.params.data_ptr(), L.ptr(self.densify_stats),
                self.views[vi].target.data_ptr(), self.rast.keys.data_ptr(), self.pos_lr, self.iterations,
                self.lr_planes[3:].tobytes(), self.lambda_dssim, tuple(self.reg))

    def _graph_eligible(self) -> bool:
        return (self.use_graph and not self.dp and not self.timer.enabled
                and (self.flame_ft is not None or self._frames_all is not None))

    def _capture_step(self, it: int):
        """Capture iteration `it` (its view, camera, buffers; the step-dependent scalars live in self._state on the device).
        With FLAME fine-tuning the graph expects the frames of this view in buffer set it&1 and leaves the next view's in the
        other one: FLAME backward + FLAME Adam + next FLAME forward fork onto the side stream under the Gaussians' Adam."""
        it0 = it
        view = self.view_for_step(it)
        cam = self._cam(view, self.sh_degree)
        r, ft, lib = self.rast, self.flame_ft, L.load()
        sched = L.LrScheduleC(float(self.pos_lr[0]), float(self.pos_lr[1]), int(self.iterations), float(self.opt.ap.beta1), float(self.opt.ap.beta2))
        def one_iteration(it):
            s = L.stream_ptr()
            L.check(lib.omfs_step_advance(L.ptr(self._state), sched, L.ptr(self._next_table), int(self._next_table.shape[0]),
                                          L.ptr(self._next_t), s), "omfs_step_advance")
            if ft is not None:
                self.dflame.slot = it & 1
                _, _, verts_all, face_all, _ = self.dflame._buffers(1)
                fxf, verts = face_all[0], verts_all[0]
            else:
                fxf, verts = self._frames_all[view.timestep], None
            g = r._gauss(self.model)
            r.project(self.model, fxf, cam)
            L.check(lib.omfs_bin_count(g, cam, r.rb, s), "omfs_bin_count")
            L.check(lib.omfs_bin_scan(cam, r.rb, s), "omfs_bin_scan")
            L.check(lib.omfs_bin_scatter(g, cam, r.rb, s), "omfs_bin_scatter")
            L.check(lib.omfs_tile_sort(cam, r.rb, s), "omfs_tile_sort")
            r.composite(cam)
            target = view.target
            if target.dtype == torch.uint8:
                if self._target_f32 is None:
                    raise RuntimeError("8-bit targets: run one eager step first")
                L.check(lib.omfs_rgb8_to_image(L.ptr(target), r.width, r.height, L.ptr(self._target_f32), s), "omfs_rgb8_to_image")
                target = self._target_f32
            r.loss_l1_ssim(target, self.lambda_dssim)
            gb = r.grad_buffers(self.grads, r.dimage, self.densify_stats, ft.dface if ft is not None else None)
            L.check(lib.omfs_composite_bwd(cam, r.rb, gb, s), "omfs_composite_bwd")
            rp = L.RegParamsC(*[float(x) for x in self.reg], L.ptr(r.n_visible))
            L.check(lib.omfs_project_bwd(g, L.ptr(fxf), cam, r.rb, gb, rp, s), "omfs_project_bwd")
            m = self.model
            self.opt.ap.grad_scale = 1.0

            def adam():
                L.check(lib.omfs_adam_step_dev(L.ptr(m.params), L.ptr(self.grads), L.ptr(self.opt.m), L.ptr(self.opt.v), m.n, m.n_pad,
                                               self.opt.ap, L.ptr(self._state), 0, NPLANES, L.stream_ptr()), "omfs_adam_step_dev")

            def flame_tail():
                # FLAME Adam, then the NEXT view's pose: its timestep comes from device memory (self._next_t, written by the
                # graph's first node from the device-resident schedule) -- with a shuffled schedule the successor of a view
                # changes from epoch to epoch
                ft.step(1.0, state_dev=L.ptr(self._state))
                self.dflame.slot = (it + 1) & 1
                self.dflame.face_frames_indexed(self._next_t)
                self.dflame.slot = it & 1

            if ft is None:
                adam()
            else:
                ft._t = view.timestep
                ft.backward(verts, 1, 0)              # the three gathers stay in front of the Adam pass (see the eager path)
                if os.environ.get("OMFS_GRAPH_FORK", "1") != "0":
                    fork, join = torch.cuda.Event(), torch.cuda.Event()
                    fork.record()
                    with torch.cuda.stream(self._side_stream):       # a parallel branch of the graph beside the Adam pass
                        self._side_stream.wait_event(fork)
                        flame_tail()
                        join.record(self._side_stream)
                    adam()
                    torch.cuda.current_stream().wait_event(join)
                else:
                    adam()
                    flame_tail()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for j in range(self.graph_iters):
                one_iteration(it0 + j)
        return graph

    def _step_graph(self, it: int) -> int:
        """Replay (capturing it on the second visit of its key) the graph of iteration `it`; returns the number of iterations
        the replay ran (graph_iters), 0: the caller steps eagerly."""
        key = self._graph_key(it)
        graph = self._graphs.get(key)
        if graph is None:
            if key not in self._graph_seen:      # first visit: eager (one-off work such as function attributes happens there)
                self._graph_seen.add(key)
                return 0
            if len(self._graphs) > 4 * max(len(self.views), 1):
                self._graphs.clear()
        view = self.view_for_step(it)
        ft = self.flame_ft
        if self._state_step != it:               # eager iterations ran in between: hand the step counts to the device
            self._state[:2].copy_(torch.tensor([self.opt.step_count, ft.step_count if ft is not None else 0], dtype=torch.int32))
        if ft is not None:
            ft.bind(self.model.binding)
            for slot in (0, 1):                  # both buffer sets exist before anything is captured (no allocation inside)
                self.dflame.slot = slot
                self.dflame._buffers(1)
            if self._frames_ready != (it, view.timestep):      # the previous iteration was not a replay: pose this view now
                torch.cuda.current_stream().wait_stream(self._side_stream)
                self.dflame.slot = it & 1
                self._pose_frames(it)
        if not (self._table_base <= it and it + self.graph_iters < self._table_base + int(self._next_table.shape[0])) or self._state_step != it:
            # (re)build the device-resident schedule: the FLAME timestep of the view of every iteration from `it` on; the graph's
            # first node reads the entry of iteration it + 1 (the view it poses last) -- nothing is copied between replays
            torch.cuda.synchronize(self.device)
            n_tab = int(self._next_table.shape[0])
            tab = [self.view_for_step(sidx).timestep for sidx in range(it, it + n_tab)]
            self._next_table.copy_(torch.tensor(tab, dtype=torch.int32))
            self._table_base = it
            self._state[L.STEP_STATE_TABLE_BASE:L.STEP_STATE_TABLE_BASE + 1].copy_(torch.tensor([it], dtype=torch.int32))
            torch.cuda.synchronize(self.device)
        if graph is None:
            self.opt.set_lr(self.lr_planes)
            graph = self._graphs[key] = self._capture_step(it)
        graph.replay()
        dbg = os.environ.get("OMFS_GRAPH_DEBUG", "")
        if dbg == "sync":
            torch.cuda.synchronize(self.device)
        elif dbg == "fence":
            self._next_t.add_(0)          # an eager kernel between two replays
        G = self.graph_iters
        self.opt.step_count += G
        if ft is not None:
            ft.step_count += G
            self._frames_ready = (it + G, self.view_for_step(it + G).timestep)
        self._state_step = it + G
        return G


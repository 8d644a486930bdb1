"""Training / rendering loops of the engine (what the absent `gaussian_avatars_repo/train.py` and
`render.py` did per iteration / per frame; call sites `02_Visual_Engine/train_ghost.py:227-271`,
`render_surgery.py:289-315`; algorithm: SURVEY.md Appendix A items 1-9).

One process per GPU.  A training step of rank r is
    FLAME(t) -> triangle frames -> project -> bin/sort -> composite -> L1+D-SSIM -> composite bwd
    -> projection bwd (+ regularisers) -> [RCCL all-reduce of the one [59][n_pad] gradient buffer]
    -> fused Adam
everything enqueued on the current HIP stream without a host sync.  Views shard across ranks
(rank r takes view `step*world + r`): the only exchange step on the path is the gradient sum.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass, field

import numpy as np
import torch

from .. import _lib as L
from .flame_rig import DeviceFlame, FlameRig
from .gaussians import GaussianModel, NPLANES, P_SH
from .graph_replay import GraphReplayMixin
from .rasterizer import Adam, Rasterizer, default_lr_planes, make_camera_struct

STAGES = ("flame", "project", "bin_count", "bin_scan", "bin_scatter", "tile_sort", "composite_fwd", "loss", "composite_bwd",
          "project_bwd", "allreduce", "adam")


@dataclass(eq=False)
class View:
    camera: dict          # synthetic.make_camera / dataset.camera_from_frame schema
    timestep: int
    target: torch.Tensor | None = None   # [3][H][W] fp32 on the device (training only)
    name: str = ""


class StageTimer:
    """HIP events between stages on the stream the kernels run on (torch's current stream is the
    stream handed to every C-ABI call).  Disabled timers cost nothing."""

    def __init__(self, enabled=False):
        self.enabled = enabled
        self.records = []   # list of (stage, start_event, end_event)
        self._last = None

    def begin(self):
        if self.enabled:
            self._last = torch.cuda.Event(enable_timing=True)
            self._last.record()

    def mark(self, stage: str):
        if self.enabled:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self.records.append((stage, self._last, e))
            self._last = e

    def summary(self) -> dict:
        """stage -> (mean ms, count); call after a device sync."""
        acc: dict = {}
        for stage, a, b in self.records:
            acc.setdefault(stage, []).append(a.elapsed_time(b))
        return {k: (float(np.mean(v)), len(v)) for k, v in acc.items()}


def expon_lr(step, lr_init, lr_final, max_steps, delay_mult=0.01, delay_steps=0):
    """3DGS `get_expon_lr_func`: log-linear interpolation lr_init -> lr_final."""
    if lr_init == lr_final or max_steps <= 0:
        return lr_init
    t = min(max(step / max_steps, 0.0), 1.0)
    return math.exp(math.log(lr_init) * (1 - t) + math.log(lr_final) * t)


def _cached_camera(cache: dict, camera: dict, sh_degree: int, bg):
    """omfs_camera struct of a camera dict, built once.  Keyed on the identity of the DICT and holding a reference to it, so
    the address cannot be handed to another object while the entry lives (a key on id(view) is not safe: callers build a
    short-lived View per frame and CPython reuses the freed addresses -- the cache then returns another frame's camera)."""
    key = (id(camera), sh_degree)
    hit = cache.get(key)
    if hit is None or hit[0] is not camera:
        hit = cache[key] = (camera, make_camera_struct(camera, sh_degree=sh_degree, bg=bg))
    return hit[1]


class _Done:
    """Stands in for the work handle of a collective that was enqueued in stream order."""

    def wait(self):
        return None


_DONE = _Done()


class Trainer(GraphReplayMixin):
    def __init__(self, rig: FlameRig, flame_params: dict, gaussians: dict, views: list, width: int, height: int,
                 bg=(0.0, 0.0, 0.0), device="cuda", iterations: int = 30000, lambda_dssim: float = 0.2,
                 reg=(0.01, 1.0, 1.0, 0.6), position_lr_init=5e-3, position_lr_final=5e-5,
                 sh_degree_max: int = 3, sh_increase_every: int = 1000, start_sh_degree: int = 0,
                 dup_capacity: int | None = None, rank: int = 0, world_size: int = 1, process_group=None,
                 n_capacity: int | None = None, finetune_flame: bool = False, flame_lr: dict | None = None,
                 coherent_order: bool = False, shuffle_views: int | None = None):
        """shuffle_views: seed of the per-epoch random view order (None: the views in index order, epoch after epoch).
        coherent_order: store the cloud along a Morton curve over the parent triangles (gaussians.coherent_order); the
        engine CLIs and the benchmark do, `model.to_dict()` returns the caller's order either way."""
        self.device = torch.device(device)
        self.rank, self.world = rank, world_size
        self.pg = process_group
        self.view_seed = None if shuffle_views is None else int(shuffle_views)
        self.dflame = DeviceFlame(rig, flame_params, device=device)
        order = None
        if coherent_order:
            from .gaussians import coherent_order as _order
            order = _order(gaussians["binding"], rig.v_template[rig.faces].mean(1))
        self.model = GaussianModel(gaussians, device=device, order=order)
        self.rast = Rasterizer(self.model.n, width, height, device=device, dup_capacity=dup_capacity, n_capacity=n_capacity)
        self.rast._ensure_bwd()
        self.rast.rb.n_visible = L.ptr(self.rast.n_visible)   # counted by omfs_bin_count (no omfs_count_visible dispatches)
        self.views = views
        self.bg = tuple(bg)
        self.iterations = iterations
        self.lambda_dssim = lambda_dssim
        self.reg = reg
        self.lr_planes = default_lr_planes(position_lr=position_lr_init)
        self.pos_lr = (position_lr_init, position_lr_final)
        self.opt = Adam(self.model, self.lr_planes)
        self.flame_ft = None           # FLAME-parameter fine-tuning (engine/flame_finetune.py)
        self._grad_head = 0            # floats in front of the planes inside grad_store (the FLAME gradients, data parallel)
        # One GPU: the gradient of the 45 SH planes of degree >= 1 is Y_k(dir) * dL/dcolour per Gaussian; project_bwd leaves the
        # six values it is made of (drgb1, dir1) and the Adam pass of those planes forms it in place (omfs_adam_step_sh_rest):
        # 2 x 54 MB of gradient traffic per iteration never happen, parameters and moments come out bit-identical.
        # OMFS_SH_ADAM=0 (or `sh_adam = False` before the first step): every plane through the gradient buffer, one Adam launch.
        self.sh_adam = world_size == 1 and process_group is None and os.environ.get("OMFS_SH_ADAM", "1") != "0"
        self.drgb1 = self.dir1 = None
        self.alloc_grads(self.model.n_pad)
        self.sh_degree_max, self.sh_every, self.sh_degree = sh_degree_max, sh_increase_every, start_sh_degree
        self.step_idx = 0
        # per-view quadrant-depth tables (omfs_raster_buffers.quad_depth): the backward's work test, and -- kept from one visit of
        # a view to the next -- the forward's hint which quadrants go deep (0.13 MB per view at 1080p)
        self._qdepth = {}
        self.timer = StageTimer(False)
        self._cams = {}
        self.densify_stats = None      # [2][n_pad] when adaptive density control is on (engine/densify.py)
        # data-parallel exchange: "compact" (default) = all-reduce of the 14 non-rank-1 planes + all-gather of dL/dcolour,
        # the 45 higher SH planes are rebuilt on every rank (engine/distributed.py); "full" = one all-reduce of all 59
        # OMFS_DP_FORCE=1 takes the exchange path with a single rank too: the RCCL calls and their stream ordering can then
        # be exercised on a one-GPU box (tests/test_gpu_distributed.py); the result is the plain single-GPU step's
        self.dp = self.world > 1 or (os.environ.get("OMFS_DP_FORCE") == "1" and process_group is not None)
        mode = os.environ.get("OMFS_DP_EXCHANGE", "compact")
        if mode not in ("compact", "full", "sharded"):
            raise ValueError(f"OMFS_DP_EXCHANGE={mode!r}: expected compact, full or sharded")
        self.compact_dp = self.dp and self.world <= 16 and mode == "compact"
        # "sharded": reduce-scatter of the gradient buffer, Adam on this rank's contiguous 1/W of the [59][n_pad] elements,
        # all-gather of the updated parameters (the Adam moments of the other shards are not maintained on this rank)
        self.sharded_dp = self.dp and mode == "sharded"
        # OMFS_DP_IMPL=abi: the "full" exchange through the C ABI's own RCCL communicator (omfs_rccl_allreduce_grads) instead of
        # torch.distributed -- the path a host without PyTorch takes; the process group then only carries the communicator's id
        self._abi_comm = None
        if self.dp and os.environ.get("OMFS_DP_IMPL", "torch") == "abi":
            if self.sharded_dp:
                raise ValueError("OMFS_DP_IMPL=abi drives the full or the compact exchange: set OMFS_DP_EXCHANGE=full or compact")
            import atexit
            from .distributed import AbiComm
            self._abi_comm = AbiComm(self.rank, self.world, process_group)
            atexit.register(self.close)          # a run that never calls close() still destroys the communicator before HIP unloads
        self._dp_patterns = {}
        self._view_steps = {}
        self._gshard = None
        self.drgb_local = self.drgb_scratch = self.drgb_all = self.cam_pos_table = None
        if self.compact_dp:
            cp = np.stack([np.asarray(make_camera_struct(v.camera).cam_pos, np.float32) for v in views])
            self.cam_pos_table = torch.from_numpy(cp).to(self.device)
        self._side_stream = torch.cuda.Stream(device=self.device)
        # FLAME Adam + the next view's pose: in stream order in front of the Gaussians' Adam pass (default), or forked onto the
        # side stream beside it (OMFS_FLAME_FORK=1).  Measured at the bench size: alone the four small kernels take 5 + 7 + 12 +
        # 5 us; beside the bandwidth-bound Adam pass 12 + 23 + 37 + 5 us, ending after it, plus ~13 us for the cross-queue
        # event to arrive: 1110 against 1101 it/s.
        self._flame_fork = os.environ.get("OMFS_FLAME_FORK", "0") == "1"
        # whole iterations as hipGraphs (one per view and buffer parity; single GPU): the step-dependent scalars -- position
        # learning rate, Adam bias corrections -- live in an omfs_step_state on the device, advanced by the graph's first node
        self.use_graph = world_size == 1 and os.environ.get("OMFS_STEP_GRAPH", "0") == "1"
        # graph_iters = G > 1 (even; single-view training only): ONE graph holds G consecutive iterations, which divides the idle
        # time ROCm leaves in front of every graph launch (~0.1 ms) by G; a `step()` that replays then advances step_idx by G.
        # A measurement switch (bench.py --graph_aux sets the attribute; there is no environment variable): engine/train.py
        # counts one iteration per step() and refuses G > 1, and a step whose G iterations would span an SH-degree change is
        # taken eagerly, one iteration at a time (see step()).
        self.graph_iters = 1
        self._graphs, self._graph_seen = {}, set()
        self._state = torch.zeros(L.STEP_STATE_WORDS, dtype=torch.int32, device=self.device)
        self._state_step, self._frames_ready = -1, None
        self._next_table = torch.zeros(4096, dtype=torch.int32, device=self.device)     # FLAME timestep of the views of the coming iterations
        self._table_base = -(1 << 30)
        self._next_t = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._events = [(torch.cuda.Event(), torch.cuda.Event()), (torch.cuda.Event(), torch.cuda.Event())]
        self._prefetch = None
        self._target_f32 = None
        if finetune_flame:
            from .flame_finetune import FlameFineTuner
            self.flame_ft = FlameFineTuner(self.dflame, flame_params, flame_lr)
            if self.dp and not self.sharded_dp:
                # data parallel: the FLAME gradients live in front of the Gaussians' gradient planes, in one allocation -- the
                # all-reduce of the planes (14 of them in the compact exchange, all 59 in the full one) carries them along
                self._grad_head = (self.flame_ft.n_grad + 63) // 64 * 64
                self.alloc_grads(self.model.n_pad)
        # A fixed FLAME sequence is posed ONCE: the triangle frames of every timestep stay resident ([T][F][16] fp32, 0.65 MB
        # per timestep at FLAME's size -- a few hundred MB of the 288 GB), and a step starts with its projection instead of
        # a cross-stream wait for the frames (15 us of idle queue per iteration in the kernel trace).  Sequences too long for
        # that, and fine-tuning (the frames change with every update), pose per step as before.
        self._frames_all = None
        T, F = self.dflame.n_frames, self.dflame.rig.n_faces
        if not finetune_flame and T * F * 64 <= int(os.environ.get("OMFS_FRAME_TABLE_BYTES", 8 << 30)):
            self._frames_all = torch.empty(T, F, 16, device=self.device)
            for t0 in range(0, T, 16):
                nb = min(16, T - t0)
                self._frames_all[t0:t0 + nb].copy_(self.dflame.face_frames(t0, nb)[1])

    def _cam(self, view: View, sh_degree: int):
        return _cached_camera(self._cams, view.camera, sh_degree, self.bg)

    def alloc_grads(self, n_pad: int) -> None:
        """(Re)allocate the gradient planes [59][n_pad] (construction, densification, rollback).  One allocation `grad_store` =
        [FLAME gradients, padded to 64 floats | 59 planes]: with data-parallel FLAME fine-tuning one all-reduce sums both."""
        self.grad_store = torch.zeros(self._grad_head + NPLANES * n_pad, device=self.device)
        self.grads = self.grad_store[self._grad_head:].view(NPLANES, n_pad)
        if self.sh_adam:
            self.drgb1 = torch.zeros(3, n_pad, device=self.device)
            self.dir1 = torch.zeros(3, n_pad, device=self.device)
        if self._grad_head and self.flame_ft is not None:
            self.flame_ft.rebind_grads(self.grad_store[:self._grad_head])

    def _frame_key(self, step: int):
        from .distributed import view_index
        if self.compact_dp:
            return (step & 1,) + tuple(self.views[view_index(step, r, self.world, len(self.views), self.view_seed)].timestep for r in range(self.world))
        return (step & 1, self.view_for_step(step).timestep)

    def _pose_frames(self, step: int):
        """FLAME forward for the frames of `step` on the current stream: (verts, face_xf, batch, own column, dp pattern)."""
        if self.compact_dp:            # pose ALL ranks' views of this step in one batch (same cost as one frame)
            pat = self._dp_pattern(step)
            verts, face_xf = self.dflame.face_frames_indexed(pat[0])
            return verts, face_xf, self.world, self.rank, pat
        verts, face_xf = self.dflame.face_frames(self.view_for_step(step).timestep, 1)
        return verts, face_xf, 1, 0, None

    def _dp_pattern(self, step: int):
        """(device int32 timesteps [W], omfs_view_set) of ALL ranks' views at this step; the schedule is periodic,
        so the few distinct patterns are built once (no per-step host-to-device copy)."""
        from .distributed import view_index
        ids = tuple(view_index(step, r, self.world, len(self.views), self.view_seed) for r in range(self.world))
        pat = self._dp_patterns.get(ids)
        if pat is None:
            ts = torch.tensor([self.views[i].timestep for i in ids], dtype=torch.int32, device=self.device)
            vs = L.ViewSetC()
            vs.n_views = self.world
            for w, i in enumerate(ids):
                vs.view[w] = i
            pat = (ts, vs, ts.to(torch.int64))
            self._dp_patterns[ids] = pat
        return pat

    def _view_step(self, view: View, cam, fxf: torch.Tensor, g, ft):
        """omfs_view_step of this view (cached with everything it points at: the structs must outlive the call)."""
        target = view.target
        # compact exchange / the in-place SH Adam of one GPU: project_bwd leaves the 45 SH planes out
        split = self.sh_adam and not self.dp
        drgb = L.ptr(self.drgb_scratch) if self.compact_dp else (L.ptr(self.drgb1) if split else 0)
        vdir = L.ptr(self.dir1) if split else 0
        key = (id(view), id(cam), fxf.data_ptr(), g.n, g.n_pad, g.params, g.binding, L.ptr(self.densify_stats), L.ptr(ft.dface) if ft is not None else 0,
               target.data_ptr(), self.lambda_dssim, tuple(self.reg), self.rast.rb.keys, self.rast.rb.dup_capacity, drgb, vdir, L.ptr(self.grads))
        hit = self._view_steps.get(key)
        if hit is None:
            if len(self._view_steps) > 8 * max(len(self.views), 1):
                self._view_steps.clear()
            r = self.rast
            gb = r.grad_buffers(self.grads, r.dimage, self.densify_stats, ft.dface if ft is not None else None)
            gb.drgb_out, gb.dir_out = drgb, vdir
            rp = L.RegParamsC(*[float(x) for x in self.reg], L.ptr(r.n_visible))
            u8 = target.dtype == torch.uint8
            if u8 and self._target_f32 is None:
                self._target_f32 = torch.empty(3, r.height, r.width, device=self.device)
            import ctypes as C
            vs = L.ViewStepC(C.pointer(g), L.ptr(fxf), C.pointer(cam), C.pointer(r.rb), C.pointer(gb), C.pointer(rp),
                             0 if u8 else L.ptr(target), L.ptr(target) if u8 else 0, L.ptr(self._target_f32) if u8 else 0,
                             float(self.lambda_dssim), L.ptr(r.loss), L.ptr(r.loss_scratch))
            hit = self._view_steps[key] = (vs, g, gb, rp, cam, fxf)
        return hit[0]

    def _allgather_drgb(self):
        """All ranks' dL/dcolour planes (compact exchange): asynchronously on torch's collective stream -- it then runs under
        omfs_project_bwd at the price of two stream hand-overs -- or, with the C ABI's own communicator, in stream order."""
        if self._abi_comm is not None:
            self._abi_comm.allgather_(self.drgb_all, self.drgb_local)
            return _DONE
        from .distributed import allgather_into_
        return allgather_into_(self.drgb_all, self.drgb_local, self.pg, async_op=True)

    def view_for_step(self, step: int) -> View:
        from .distributed import view_index
        return self.views[view_index(step, self.rank, self.world, len(self.views), self.view_seed)]

    def invalidate_graphs(self) -> None:
        """Buffers a captured iteration or a cached omfs_view_step refers to were replaced (densification): forget them all."""
        self._graphs.clear()
        self._graph_seen.clear()
        self._view_steps.clear()

    def sync_optimizer_state(self) -> None:
        """"sharded" exchange only: a rank maintains Adam's moments for its own 1/W of the flat [59 * n_pad] range; before anything
        reads them WHOLE (a checkpoint, the compaction of a densification, whose new n_pad also moves every shard boundary)
        the shards are all-gathered into every rank's full buffers.  A no-op in every other mode."""
        if not self.sharded_dp:
            return
        from .distributed import allgather_shards_
        allgather_shards_(self.opt.m.view(-1), self.pg)
        allgather_shards_(self.opt.v.view(-1), self.pg)

    def step(self) -> None:
        """One training iteration of this rank (enqueue only)."""
        it = self.step_idx
        view = self.view_for_step(it)
        if it > 0 and it % self.sh_every == 0 and self.sh_degree < self.sh_degree_max:
            self.sh_degree += 1
        G = self.graph_iters
        if G > 1 and (G % 2 or len(self.views) != 1):
            raise ValueError(f"graph_iters={G}: several iterations per graph need an even count (two FLAME buffer sets alternate) and a single view")
        # a replay of G iterations must not run past the iteration that raises the SH degree (the graph holds ONE degree): the
        # steps in front of such a boundary are taken eagerly, one at a time, so `it % sh_every == 0` is seen above
        spans_sh_change = G > 1 and self.sh_degree < self.sh_degree_max and (it // self.sh_every) != ((it + G - 1) // self.sh_every)
        qd = self._qdepth.get(id(view))
        if qd is None:
            qd = self._qdepth[id(view)] = torch.zeros(self.rast.n_tiles, 4, dtype=torch.int32, device=self.device)
        self.rast.rb.quad_depth = L.ptr(qd)
        if self._graph_eligible() and not spans_sh_change:
            try:
                done = self._step_graph(it)
                if done:
                    self.step_idx += done
                    return
            except Exception as e:      # a failed capture must not cost the run: say so, go on eagerly
                print(f"[engine] hipGraph capture of the training iteration failed ({type(e).__name__}: {e}); continuing without graphs")
                self.use_graph = False
                self._graphs.clear()
                torch.cuda.synchronize()
        cam = self._cam(view, self.sh_degree)
        r, tm = self.rast, self.timer
        tm.begin()
        ft = self.flame_ft
        # FLAME fine-tuning on one GPU: the FLAME backward chain, the FLAME parameters' Adam and the NEXT step's FLAME forward
        # run on the side stream underneath the Gaussians' Adam pass (nothing of it touches the Gaussian parameters); the
        # posed buffers alternate between two sets
        ft_pipe = ft is not None and not self.dp
        if ft is not None:
            ft.begin(view.timestep, self.model.binding, all_timesteps=self.compact_dp)
            if ft_pipe and self._prefetch is not None and self._prefetch[0] == (it, view.timestep):
                if self._prefetch[1] is not None:
                    torch.cuda.current_stream().wait_event(self._prefetch[1])
                self.dflame.slot = it & 1
                verts, face_xf, nb, col, pat = self._prefetch[2]
            else:
                if ft_pipe:
                    self.dflame.slot = it & 1
                verts, face_xf, nb, col, pat = self._pose_frames(it)
        elif self._frames_all is not None:
            # the sequence is fixed and its frames are resident: nothing to pose
            if self.compact_dp:            # the frames of ALL ranks' views of this step, one row per rank
                pat = self._dp_pattern(it)
                face_xf, nb, col, verts = self._frames_all.index_select(0, pat[2]), self.world, self.rank, None
            else:
                t = view.timestep
                face_xf, nb, col, verts, pat = self._frames_all[t:t + 1], 1, 0, None, None
        else:
            # the sequence is fixed: this step's frames were posed on the side stream during the previous step,
            # the next step's are posed now, concurrently with this step's kernels (double-buffered outputs)
            slot = it & 1
            want = self._frame_key(it)
            if self._prefetch is not None and self._prefetch[0] == want:
                torch.cuda.current_stream().wait_event(self._prefetch[1])
                verts, face_xf, nb, col, pat = self._prefetch[2]
            else:
                self.dflame.slot = slot
                verts, face_xf, nb, col, pat = self._pose_frames(it)
            ev_main, ev_side = self._events[slot]
            ev_main.record()
            with torch.cuda.stream(self._side_stream):
                self._side_stream.wait_event(ev_main)       # the other buffer set was last read by the previous step
                self.dflame.slot = slot ^ 1
                nxt = self._pose_frames(it + 1)
                ev_side.record(self._side_stream)
            self._prefetch = (self._frame_key(it + 1), ev_side, nxt)
            self.dflame.slot = slot
        tm.mark("flame")
        fxf = face_xf[col]
        gather = None
        if self.compact_dp and (self.drgb_local is None or self.drgb_local.shape[1] != self.model.n_pad):
            self.drgb_local = torch.zeros(3, self.model.n_pad, device=self.device)
            self.drgb_scratch = torch.zeros(3, self.model.n_pad, device=self.device)
            self.drgb_all = torch.zeros(self.world, 3, self.model.n_pad, device=self.device)
        if not tm.enabled and not self.dp:
            # one C-ABI call for the whole view (projection ... parameter gradients): a Python host pays ~10 us per ctypes call
            lib, s, g = L.load(), L.stream_ptr(), r._gauss(self.model)
            L.check(lib.omfs_view_forward_backward(self._view_step(view, cam, fxf, g, ft), s), "omfs_view_forward_backward")
        elif not tm.enabled:
            # data parallel: the same call in its two halves (ABI 7) -- projection ... composite_bwd (+ the rank's dL/dcolour
            # planes), then project_bwd, with the all-gather of dL/dcolour issued between them so that it runs under project_bwd
            lib, s, g = L.load(), L.stream_ptr(), r._gauss(self.model)
            vs = self._view_step(view, cam, fxf, g, ft)
            L.check(lib.omfs_view_forward_composite_bwd(vs, L.ptr(self.drgb_local) if self.compact_dp else 0, s), "omfs_view_forward_composite_bwd")
            if self.compact_dp:
                from .distributed import allgather_into_
                gather = self._allgather_drgb()
            L.check(lib.omfs_view_project_bwd(vs, s), "omfs_view_project_bwd")
        else:
            r.project(self.model, fxf, cam); tm.mark("project")
            lib = L.load()
            g = r._gauss(self.model)
            s = L.stream_ptr()
            L.check(lib.omfs_bin_count(g, cam, r.rb, s), "omfs_bin_count"); tm.mark("bin_count")
            L.check(lib.omfs_bin_scan(cam, r.rb, s), "omfs_bin_scan"); tm.mark("bin_scan")
            L.check(lib.omfs_bin_scatter(g, cam, r.rb, s), "omfs_bin_scatter"); tm.mark("bin_scatter")
            L.check(lib.omfs_tile_sort(cam, r.rb, s), "omfs_tile_sort"); tm.mark("tile_sort")
            r.composite(cam); tm.mark("composite_fwd")
            target = view.target
            if target.dtype == torch.uint8:        # 8-bit [H][W][3] targets (large datasets): expanded per view on the device
                if self._target_f32 is None:
                    self._target_f32 = torch.empty(3, r.height, r.width, device=self.device)
                L.check(lib.omfs_rgb8_to_image(L.ptr(target), r.width, r.height, L.ptr(self._target_f32), s), "omfs_rgb8_to_image")
                target = self._target_f32
            r.loss_l1_ssim(target, self.lambda_dssim); tm.mark("loss")
            split = self.sh_adam and not self.dp
            gb = r.grad_buffers(self.grads, r.dimage, self.densify_stats, ft.dface if ft is not None else None,
                                self.drgb_scratch if self.compact_dp else (self.drgb1 if split else None), self.dir1 if split else None)
            L.check(lib.omfs_composite_bwd(cam, r.rb, gb, s), "omfs_composite_bwd"); tm.mark("composite_bwd")
            gather = None
            if self.compact_dp:            # dL/dcolour is final here: its all-gather runs under project_bwd
                from .distributed import allgather_into_
                L.check(lib.omfs_extract_drgb(r.rb, L.ptr(r.dsplat), self.model.n, self.model.n_pad, L.ptr(self.drgb_local), s), "omfs_extract_drgb")
                gather = self._allgather_drgb()
            rp = L.RegParamsC(*[float(x) for x in self.reg], L.ptr(r.n_visible))
            L.check(lib.omfs_project_bwd(g, L.ptr(fxf), cam, r.rb, gb, rp, s), "omfs_project_bwd"); tm.mark("project_bwd")
        if ft_pipe:
            # The three gathers of the FLAME backward always stay on this stream: beside the bandwidth-bound Adam pass they run
            # 3-5x slower (measured: 34 + 28 + 50 us instead of 12 + 8 + 15).  The FLAME Adam and the next view's pose (joints,
            # LBS, triangle frames) follow in stream order, or fork with OMFS_FLAME_FORK=1 (see __init__).
            ft.backward(verts[col], nb, col)
            if self._flame_fork:
                ev_main, ev_side = self._events[it & 1]
                ev_main.record()
                with torch.cuda.stream(self._side_stream):
                    self._side_stream.wait_event(ev_main)
                    ft.step(1.0)
                    nview = self.view_for_step(it + 1)
                    self.dflame.slot = (it + 1) & 1
                    nxt = self._pose_frames(it + 1)
                    ev_side.record(self._side_stream)
                self._prefetch = ((it + 1, nview.timestep), ev_side, nxt)
            else:                                   # the same chain in stream order, in front of the Adam pass
                ft.step(1.0)
                nview = self.view_for_step(it + 1)
                self.dflame.slot = (it + 1) & 1
                nxt = self._pose_frames(it + 1)
                self._prefetch = ((it + 1, nview.timestep), None, nxt)
            self.dflame.slot = it & 1
            tm.mark("flame_bwd")
        elif ft is not None:
            ft.backward(verts[col], nb, col); tm.mark("flame_bwd")
        if self.dp:
            from .distributed import allgather_into_, allreduce_sum_
            if self.compact_dp:
                # 11 contiguous planes, asynchronously: the 48 rebuilt SH planes are updated while they are on the links
                # (with FLAME fine-tuning: the FLAME gradients in front of plane 0 travel in the same collective)
                low = self.grad_store[:self._grad_head + P_SH * self.model.n_pad]     # 11 planes: xyz, scale, rotation, opacity
                if self._abi_comm is not None:      # in stream order on the compute stream: no hand-over, no overlap either
                    self._abi_comm.allreduce_(low)
                    reduce14 = _DONE
                else:
                    reduce14 = allreduce_sum_(low, self.pg, async_op=True)
                gather.wait()
                L.check(lib.omfs_sh_rest_grads(g, L.ptr(face_xf), self.dflame.rig.n_faces, L.ptr(self.cam_pos_table), pat[1],
                                               L.ptr(self.drgb_all), self.sh_degree, L.ptr(self.grads), s), "omfs_sh_rest_grads")
            elif self.sharded_dp:
                from .distributed import reduce_scatter_sum_
                flat = self.grads.view(-1)
                S = flat.numel() // self.world            # n_pad is a multiple of 256: every world size up to 64 divides 59*n_pad/4
                if self._gshard is None or self._gshard.numel() != S:
                    if flat.numel() % (4 * self.world):
                        raise ValueError("sharded exchange needs 59 * n_pad divisible by 4 * world_size")
                    self._gshard = torch.empty(S, device=self.device)
                reduce_scatter_sum_(self._gshard, flat, self.pg)
            elif self._abi_comm is not None:
                self._abi_comm.allreduce_(self.grad_store)    # the same collective, issued by the library (omfs_rccl_allreduce_grads)
            else:
                allreduce_sum_(self.grad_store, self.pg)      # all 59 planes (+ the FLAME gradients in front of them)
            if ft is not None and self.sharded_dp:       # every rank touched a different timestep: dense (tiny) tensors, summed
                allreduce_sum_(ft.grad_flat, self.pg)          # expr, pose, translation gradients: one buffer
            tm.mark("allreduce")
        lr = expon_lr(it, self.pos_lr[0], self.pos_lr[1], self.iterations)
        self.lr_planes[0:3] = lr
        self.opt.set_lr(self.lr_planes)
        if self.compact_dp:
            self.opt.begin_step(1.0 / self.world)
            self.opt.apply_planes(self.grads, P_SH, NPLANES - P_SH)     # the 48 rebuilt SH planes
            reduce14.wait()
            self.opt.apply_planes(self.grads, 0, P_SH)
        elif self.sharded_dp:
            from .distributed import allgather_shards_
            S = self._gshard.numel()
            self.opt.begin_step(1.0 / self.world)
            self.opt.apply_range(self._gshard, self.rank * S, S)
            allgather_shards_(self.model.params.view(-1), self.pg)
        elif self.sh_adam and not self.dp:
            # ONE launch: planes 0..13 (geometry, opacity, SH degree 0) from the gradient planes, the other 45 with the gradient
            # formed in place
            self.opt.begin_step(1.0)
            self.opt.apply_sh_rest(self.drgb1, self.dir1, self.sh_degree, grads_low=self.grads)
        else:
            self.opt.step(self.grads, 1.0 / self.world)
        if ft is not None and not ft_pipe:
            ft.step(1.0 / self.world)
        tm.mark("adam")
        self.step_idx += 1

    def loss_value(self) -> float:
        """Host sync: loss of the last step on this rank."""
        return float(self.rast.loss.item())

    def close(self) -> None:
        """Release what the process would otherwise only drop at exit: the C ABI's own RCCL communicator (OMFS_DP_IMPL=abi)."""
        comm, self._abi_comm = self._abi_comm, None
        if comm is not None:
            comm.close()


class Renderer:
    """Per-frame rendering of a (possibly FLAME-edited) sequence: render_surgery's inner loop."""

    def __init__(self, rig: FlameRig, flame_params: dict, gaussians: dict, width: int, height: int,
                 bg=(0.0, 0.0, 0.0), device="cuda", sh_degree: int = 3, dup_capacity: int | None = None,
                 flame_batch: int = 16, coherent_order: bool = False, n_streams: int = 1):
        """n_streams > 1: `render_async` / `render_png_stream` deal consecutive frames to that many HIP streams, each with
        raster buffers of its own -- the frames of a sequence are independent, so the latency-bound kernels of one frame
        (binning chain, the deep forward's tail) run under the instruction-bound ones of its neighbours (2303 -> 2840 -> 3053
        frames/s with 1 / 2 / 3 streams at 1080p, 300 k Gaussians).  `render` stays on the caller's stream with the first set."""
        self.device = torch.device(device)
        self.dflame = DeviceFlame(rig, flame_params, device=device)
        order = None
        if coherent_order:
            from .gaussians import coherent_order as _order
            order = _order(gaussians["binding"], rig.v_template[rig.faces].mean(1))
        self.model = GaussianModel(gaussians, device=device, order=order)
        self.n_streams = max(1, int(n_streams))
        self.rasts = [Rasterizer(self.model.n, width, height, device=device, dup_capacity=dup_capacity) for _ in range(self.n_streams)]
        for r in self.rasts:
            r.rb.flags = L.RB_FORWARD_ONLY     # no backward pass follows: the forward skips the segment checkpoints
        self.rast = self.rasts[0]
        self._streams = [torch.cuda.Stream(device=self.device) for _ in range(self.n_streams)] if self.n_streams > 1 else []
        self._slot_done = [None] * self.n_streams      # event behind the last frame of each stream
        self._render_done = None                       # event behind the last frame of render() (caller's stream)
        self._next_slot = 0
        # the raster buffers were zero-filled on the constructor's stream and are used on the side streams: drain it once
        torch.cuda.current_stream(self.device).synchronize()
        self.bg, self.sh_degree = tuple(bg), sh_degree
        self.timer = StageTimer(False)
        self._cams = {}
        self.flame_batch = int(flame_batch)
        self._frames = None            # (first timestep, count, face_xf [count][F][16]) of the last posed batch

    def render_png_rows_to_host(self, view: View, n_slots: int = 4):
        """Enqueue one frame and its asynchronous copy to a pinned host buffer as PNG scanlines ([H][1+3W] uint8).
        Returns (numpy view of the pinned buffer, event): wait for the event (`event.synchronize()`, e.g. in the encoder
        thread) before reading; the buffer is reused after n_slots further calls."""
        self.render(view)
        rows = self.rast.to_png_rows()
        if getattr(self, "_host_ring", None) is None or len(self._host_ring) != n_slots:
            self._host_ring = [torch.empty(rows.shape, dtype=torch.uint8, pin_memory=True) for _ in range(n_slots)]
            self._host_events = [torch.cuda.Event() for _ in range(n_slots)]
            self._host_next = 0
        k = self._host_next
        self._host_next = (k + 1) % n_slots
        self._host_ring[k].copy_(rows, non_blocking=True)
        self._host_events[k].record()
        return self._host_ring[k].numpy(), self._host_events[k]

    def render_png_stream(self, view: View, n_slots: int = 4):
        """Enqueue one frame and its DEVICE-side PNG deflate (`Rasterizer.to_png_stream`) into slot k of a ring of n_slots
        stream buffers.  Returns (k, event): once the event has fired, `fetch_png_stream(k, event)` -- any thread -- brings the
        zlib stream to the host.  The slot is reused after n_slots further calls."""
        if self.n_streams > 1:
            return self._render_png_stream_pipelined(view, n_slots)
        self.render(view)
        r = self.rast
        return self._png_stream_of_frame(r, n_slots)

    def _render_png_stream_pipelined(self, view: View, n_slots: int):
        slot, stream = self._begin_async(view)
        with torch.cuda.stream(stream):
            r = self.rasts[slot]
            self._enqueue_frame(view, r)
            out = self._png_stream_of_frame(r, n_slots)
            self._end_async(slot)
        return out

    def _png_stream_of_frame(self, r, n_slots: int):
        """Device deflate of the frame in `r` into the next ring slot + its speculative copy to the host (current stream)."""
        if getattr(self, "_png_ring", None) is None or len(self._png_ring) != n_slots:
            r.to_png_stream()                              # sizes the scratch buffers
            cap = r.png_stream_capacity
            # one allocation per slot: [stream length (4 bytes) | zlib stream]: the length travels with the data in ONE copy
            self._png_ring = [torch.zeros(16 + cap, dtype=torch.uint8, device=self.device) for _ in range(n_slots)]
            self._png_host = [torch.empty(16 + cap, dtype=torch.uint8, pin_memory=True) for _ in range(n_slots)]
            self._png_host_mv = [memoryview(h.numpy()) for h in self._png_host]
            self._png_events = [torch.cuda.Event() for _ in range(n_slots)]
            self._png_done = [torch.cuda.Event() for _ in range(n_slots)]      # rebuilt WITH the ring (a larger n_slots later)
            self._png_got = [0] * n_slots
            self._png_next = 0
            self._png_guess = cap // 4
            import threading
            self._png_tls = threading.local()
            # the ring is zero-filled on THIS stream and its slots are written by deflate launches on the renderer's other
            # streams: a fill that lands late would wipe a frame's length word -- drain the filling stream once
            torch.cuda.current_stream(self.device).synchronize()
        k = self._png_next
        self._png_next = (k + 1) % n_slots
        buf = self._png_ring[k]
        r.to_png_stream(buf[16:], buf[:4].view(torch.int32))
        # the speculative device-to-host copy (length word + as many bytes as the last frames needed, + 12 %) is enqueued HERE, on
        # a copy stream behind the deflate: the encoder threads then only wait for an event -- no stream operations of their own
        # (synchronising a stream from many threads contends with this thread's launches inside the HIP runtime)
        if getattr(self, "_png_copy_stream", None) is None:
            self._png_copy_stream = torch.cuda.Stream(device=self.device)
        self._png_events[k].record()
        got = min(16 + self._png_guess, buf.numel())
        with torch.cuda.stream(self._png_copy_stream):
            self._png_copy_stream.wait_event(self._png_events[k])
            self._png_host[k][:got].copy_(buf[:got], non_blocking=True)
            self._png_done[k].record()
        self._png_got[k] = got
        return k, self._png_done[k]

    def fetch_png_stream(self, k: int, event) -> memoryview:
        """The zlib stream of ring slot k on the host (pinned memory, valid until the slot is reused).  Its length is only known
        on the device, so the copy is speculative: the length word and as many bytes as the last frames needed (+ 12 %) in one
        transfer, the rest -- rarely -- in a second one; about a fifth of the raw scanlines for a head on a plain background.
        Runs on a copy stream of the calling thread, so encoder threads do not serialise behind the render stream."""
        event.synchronize()                                   # the speculative copy has landed
        host, got = self._png_host[k], self._png_got[k]
        n = int.from_bytes(self._png_host_mv[k][:4], "little")
        if n <= 0 or 16 + n > host.numel():
            raise L.OmfsError(f"device PNG deflate reported an impossible stream length {n}")
        if 16 + n > got:                                      # rare: the frame needed more than the guess -- fetch the remainder
            tls = self._png_tls
            if getattr(tls, "stream", None) is None:
                tls.stream = torch.cuda.Stream(device=self.device)
            m = int(L.load().omfs_png_fetch(host.data_ptr(), self._png_ring[k].data_ptr(), host.numel(), n, tls.stream.cuda_stream, 0))
            if m != n:
                raise L.OmfsError(f"omfs_png_fetch failed ({m}): {L.load().omfs_last_error().decode()}")
        self._png_guess = max(int(n * 1.12) + 4096, (self._png_guess * 7) // 8)
        return self._png_host_mv[k][16:16 + n]

    def render(self, view: View, rgb8: bool = False):
        """Enqueue one frame on the current stream; returns the reused image tensor ([3][H][W] fp32 or [H][W][3] uint8)."""
        if self.n_streams > 1 and any(e is not None for e in self._slot_done):
            for e in self._slot_done:                      # frames of render_async still in flight share the posed FLAME batch
                if e is not None:
                    torch.cuda.current_stream().wait_event(e)
        self._enqueue_frame(view, self.rast)
        out = self.rast.to_rgb8() if rgb8 else self.rast.image
        if self.n_streams > 1:          # a later render_async that poses a new FLAME batch must not overwrite the frames this reads
            self._render_done = torch.cuda.Event()
            self._render_done.record()
        return out

    def render_async(self, view: View, rgb8: bool = True):
        """Enqueue one frame on the next of the renderer's streams (n_streams > 1).  Returns (tensor, event): the tensor -- that
        stream's own output buffer -- is valid once the event has fired and until the n_streams-th next call."""
        if self.n_streams == 1:
            out = self.render(view, rgb8)
            ev = torch.cuda.Event(); ev.record()
            return out, ev
        slot, stream = self._begin_async(view)
        with torch.cuda.stream(stream):
            r = self.rasts[slot]
            self._enqueue_frame(view, r)
            out = r.to_rgb8() if rgb8 else r.image
            ev = self._end_async(slot)
        return out, ev

    def _begin_async(self, view: View):
        slot = self._next_slot
        self._next_slot = (slot + 1) % self.n_streams
        stream = self._streams[slot]
        t = view.timestep
        fr = self._frames
        if fr is None or not (fr[0] <= t < fr[0] + fr[1]):
            # the next FLAME batch is posed on this frame's stream, once every stream has finished with the previous batch's
            # triangle frames (the buffers are reused); the other streams wait for the event recorded behind the pose
            for e in (*self._slot_done, self._render_done):
                if e is not None:
                    stream.wait_event(e)
            with torch.cuda.stream(stream):
                self._pose_batch(t)
        elif fr[3] is not None:
            stream.wait_event(fr[3])
        return slot, stream

    def _end_async(self, slot: int):
        ev = torch.cuda.Event()
        ev.record()
        self._slot_done[slot] = ev
        return ev

    def synchronize(self):
        """Host sync: every stream of the renderer has drained."""
        for st in self._streams:
            st.synchronize()
        torch.cuda.current_stream().synchronize()

    def check_status(self):
        for r in self.rasts:
            r.check_status()

    def overflowed(self) -> bool:
        """Host sync: did any stream's tile lists exceed the pair capacity since the flags were last cleared?  (Frames rendered
        from then on hold empty lists.)"""
        self.synchronize()
        return any(r.overflowed() for r in self.rasts)

    def grow_dup_capacity(self, factor: float = 2.0) -> int:
        """Every stream's pair-sized buffers `factor` times larger, flags cleared: the caller renders the affected frames again."""
        self.synchronize()
        return max(r.grow_dup_capacity(factor) for r in self.rasts)

    def _pose_batch(self, t: int):
        nb = max(1, min(self.flame_batch, self.dflame.n_frames - t))
        fxf = self.dflame.face_frames(t, nb)[1]
        ev = None
        if self.n_streams > 1:
            ev = torch.cuda.Event()
            ev.record()
        self._frames = (t, nb, fxf, ev)
        return self._frames

    def _enqueue_frame(self, view: View, r):
        """Project, bin, sort and composite one frame into the raster buffers `r` on the current stream."""
        cam = _cached_camera(self._cams, view.camera, self.sh_degree, self.bg)
        tm = self.timer
        tm.begin()
        # a sequence is rendered in order: FLAME is posed for `flame_batch` consecutive timesteps at once (the pass reads
        # the whole basis whatever the batch), later frames of the batch find their triangle frames ready
        t = view.timestep
        fr = self._frames
        if fr is None or not (fr[0] <= t < fr[0] + fr[1]):
            fr = self._pose_batch(t)
        fxf = fr[2][t - fr[0]]
        tm.mark("flame")
        lib = L.load()
        g = r._gauss(self.model)
        s = L.stream_ptr()
        r.project(self.model, fxf, cam); tm.mark("project")
        L.check(lib.omfs_bin_count(g, cam, r.rb, s), "omfs_bin_count"); tm.mark("bin_count")
        L.check(lib.omfs_bin_scan(cam, r.rb, s), "omfs_bin_scan"); tm.mark("bin_scan")
        L.check(lib.omfs_bin_scatter(g, cam, r.rb, s), "omfs_bin_scatter"); tm.mark("bin_scatter")
        L.check(lib.omfs_tile_sort(cam, r.rb, s), "omfs_tile_sort"); tm.mark("tile_sort")
        r.composite(cam); tm.mark("composite_fwd")

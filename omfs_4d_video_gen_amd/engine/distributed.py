"""Data-parallel plumbing: how units shard across ranks and the one exchange step of the path.

train_ghost: rank r of W takes view (step*W + r) mod n_views; after the local backward every rank
holds a full [59][n_pad] gradient buffer; the exchange is either ONE all-reduce (sum) over that buffer ("full") or,
by default, the compact form (trainer.py): all-reduce of the 11 planes that are not rank-1 (geometry, opacity) + all-gather of
each rank's dL/dcolour (3 planes), from which every rank rebuilds the summed gradient of all 48 SH planes itself
(omfs_sh_rest_grads; degree 0 too since round 5: its basis is a constant) -- 2.4x fewer bytes per rank on the xGMI links at 8 ranks.  Then every
rank applies the same Adam step with grad_scale = 1/W, so replicas stay bit-identical.  A third form ("sharded",
SURVEY.md section 8e): reduce-scatter of the whole buffer, Adam on the rank's 1/W of the elements (omfs_adam_step_range), all-gather
of the updated parameters -- the bytes of "full" on the links, Adam traffic divided by W.
render_surgery: frame f belongs to rank f mod W; no collective.
`torch.distributed` backend "nccl" is RCCL on ROCm; tests run the same code over gloo on CPU tensors.
"""
from __future__ import annotations

import functools
import os

import numpy as np
import torch
import torch.distributed as dist


@functools.lru_cache(maxsize=64)
def _epoch_order(seed: int, epoch: int, n_views: int):
    return tuple(int(i) for i in np.random.default_rng([int(seed), int(epoch)]).permutation(n_views))


def view_index(step: int, rank: int, world: int, n_views: int, seed: int | None = None) -> int:
    """View of rank `rank` at iteration `step`.  seed None: global slot step*world + rank walks the views in index order.
    Otherwise rank r walks ITS views (`views_of_rank`) epoch by epoch, every epoch in its own seeded random order (without
    replacement, as upstream pops its viewpoint stack) -- a function of (seed, rank, step) alone, so every rank can compute
    every other rank's view (the compact exchange poses them all) without talking."""
    if seed is None:
        return (step * world + rank) % n_views
    own = views_of_rank(n_views, rank, world)
    epoch, k = divmod(step, len(own))
    return own[_epoch_order(seed * 1000003 + rank, epoch, len(own))[k]]


def views_of_rank(n_views: int, rank: int, world: int) -> range:
    """The views rank `rank` ever trains on: index = rank (mod world).  A rank keeps only their images in HBM."""
    return range(rank % max(n_views, 1), n_views, world) if world <= n_views else range(rank % n_views, rank % n_views + 1)


def frames_of_rank(n_frames: int, rank: int, world: int) -> range:
    return range(rank, n_frames, world)


class AbiComm:
    """The C ABI's own RCCL communicator (`omfs_comm_*`, `omfs_rccl_*`: include/omfs_splat.h) -- what a host that is not
    PyTorch would hold.  Rank 0 draws the 128-byte id, `group` (any torch.distributed backend) only carries it to the other
    ranks; with one rank nothing is exchanged at all.  Every collective is enqueued on torch's current stream."""

    def __init__(self, rank: int = 0, world: int = 1, group=None):
        import ctypes as C
        from .. import _lib as L
        self._L, self.rank, self.world = L, int(rank), int(world)
        ident = (C.c_ubyte * 128)()
        if self.rank == 0:
            L.check(L.load().omfs_comm_unique_id(C.cast(ident, C.c_void_p)), "omfs_comm_unique_id")
        if self.world > 1:
            box = [bytes(ident)]
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            ident = (C.c_ubyte * 128).from_buffer_copy(box[0])
        self._comm = C.c_void_p()
        L.check(L.load().omfs_comm_create(C.cast(ident, C.c_void_p), self.rank, self.world, C.byref(self._comm)), "omfs_comm_create")

    def allreduce_(self, buf: torch.Tensor) -> torch.Tensor:
        L = self._L
        L.check(L.load().omfs_rccl_allreduce_grads(self._comm, L.ptr(buf), buf.numel(), L.stream_ptr()), "omfs_rccl_allreduce_grads")
        return buf

    def allgather_(self, out: torch.Tensor, mine: torch.Tensor) -> torch.Tensor:
        L = self._L
        if out.numel() != self.world * mine.numel():
            raise ValueError("out must hold world_size pieces of mine's size")
        L.check(L.load().omfs_rccl_allgather(self._comm, L.ptr(mine), L.ptr(out), mine.numel(), L.stream_ptr()), "omfs_rccl_allgather")
        return out

    def reduce_scatter_(self, shard: torch.Tensor, full: torch.Tensor) -> torch.Tensor:
        L = self._L
        if full.numel() != self.world * shard.numel():
            raise ValueError("full must hold world_size shards of shard's size")
        L.check(L.load().omfs_rccl_reduce_scatter(self._comm, L.ptr(full), L.ptr(shard), shard.numel(), L.stream_ptr()),
                "omfs_rccl_reduce_scatter")
        return shard

    def close(self):
        if self._comm:
            torch.cuda.synchronize()
            self._L.check(self._L.load().omfs_comm_destroy(self._comm), "omfs_comm_destroy")
            self._comm = None


def allreduce_sum_(buf: torch.Tensor, group=None, async_op: bool = False):
    """In-place sum over ranks of the gradient SoA (or a contiguous plane range of it).  With async_op the work handle
    is returned (call .wait() before reading buf)."""
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or os.environ.get("OMFS_DP_FORCE") == "1"):
        work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        if async_op:
            return work
    return buf


def allgather_into_(out: torch.Tensor, inp: torch.Tensor, group=None, async_op: bool = False):
    """out [W][...] <- every rank's inp [...] (rank order).  One collective; falls back to the list form where the
    backend lacks the tensor form.  With async_op the work handle is returned (call .wait() before reading out)."""
    world = dist.get_world_size(group)
    if out.shape[0] != world or out[0].shape != inp.shape:
        raise ValueError("out must be [world_size, *inp.shape]")
    try:
        work = dist.all_gather_into_tensor(out, inp, group=group, async_op=async_op)
    except (RuntimeError, NotImplementedError):
        work = dist.all_gather(list(out.unbind(0)), inp, group=group, async_op=async_op)
    return work if async_op else out


def reduce_scatter_sum_(out: torch.Tensor, full: torch.Tensor, group=None):
    """out [S] <- this rank's contiguous shard of the sum over ranks of full [W*S] (rank r owns [r*S, (r+1)*S)).
    RCCL: ONE reduce-scatter.  gloo has no reduce-scatter: the CPU rehearsal all-reduces and slices -- the same numbers
    (a ring reduce-scatter and a ring all-reduce add in the same order on every shard only by luck, so bit-equality between
    the backends is not claimed; between the RANKS of one run the result is bit-identical by construction, each element
    being updated by exactly one rank)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if full.numel() != world * out.numel():
        raise ValueError("full must hold world_size shards of out's size")
    if dist.get_backend(group) == "nccl":
        dist.reduce_scatter_tensor(out, full, op=dist.ReduceOp.SUM, group=group)
    else:
        dist.all_reduce(full, op=dist.ReduceOp.SUM, group=group)
        out.copy_(full[rank * out.numel():(rank + 1) * out.numel()])
    return out


def allgather_shards_(full: torch.Tensor, group=None):
    """full [W*S]: every rank's own shard [r*S, (r+1)*S) is broadcast to all (in place on RCCL)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    S = full.numel() // world
    mine = full[rank * S:(rank + 1) * S]
    if dist.get_backend(group) == "nccl":
        dist.all_gather_into_tensor(full, mine, group=group)
    else:
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine.clone(), group=group)
        for r, p in enumerate(parts):
            full[r * S:(r + 1) * S].copy_(p)
    return full


def replicas_in_sync(params: torch.Tensor, group=None) -> bool:
    """True when every rank holds bit-identical parameters (cheap checksum exchange)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return True
    chk = torch.stack([params.double().sum(), params.double().abs().sum()])
    lo, hi = chk.clone(), chk.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    return bool(torch.equal(lo, hi))

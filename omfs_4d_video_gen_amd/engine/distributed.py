"""Data-parallel plumbing: how units shard across ranks and the one exchange step of the path.

train_ghost: rank r of W takes view (step*W + r) mod n_views; after the local backward every rank
holds a full [59][n_pad] gradient buffer; the exchange is either ONE all-reduce (sum) over that buffer ("full") or,
by default, the compact form (trainer.py): all-reduce of the 14 planes that are not rank-1 (geometry, opacity, SH
degree 0) + all-gather of each rank's dL/dcolour (3 planes), from which every rank rebuilds the summed gradient of the
45 higher SH planes itself (omfs_sh_rest_grads) -- 2.3x fewer bytes per rank on the xGMI links at 8 ranks.  Then every
rank applies the same Adam step with grad_scale = 1/W, so replicas stay bit-identical.
render_surgery: frame f belongs to rank f mod W; no collective.
`torch.distributed` backend "nccl" is RCCL on ROCm; tests run the same code over gloo on CPU tensors.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def view_index(step: int, rank: int, world: int, n_views: int) -> int:
    return (step * world + rank) % n_views


def frames_of_rank(n_frames: int, rank: int, world: int) -> range:
    return range(rank, n_frames, world)


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


def replicas_in_sync(params: torch.Tensor, group=None) -> bool:
    """True when every rank holds bit-identical parameters (cheap checksum exchange)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return True
    chk = torch.stack([params.double().sum(), params.double().abs().sum()])
    lo, hi = chk.clone(), chk.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    return bool(torch.equal(lo, hi))

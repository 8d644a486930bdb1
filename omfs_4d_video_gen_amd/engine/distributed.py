"""Data-parallel plumbing: how units shard across ranks and the one exchange step of the path.

train_ghost: rank r of W takes view (step*W + r) mod n_views; after the local backward every rank
holds a full [59][n_pad] gradient buffer; ONE all-reduce (sum) over that buffer, then every rank
applies the same Adam step with grad_scale = 1/W, so replicas stay bit-identical.
render_surgery: frame f belongs to rank f mod W; no collective.
`torch.distributed` backend "nccl" is RCCL on ROCm; tests run the same code over gloo on CPU tensors.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def view_index(step: int, rank: int, world: int, n_views: int) -> int:
    return (step * world + rank) % n_views


def frames_of_rank(n_frames: int, rank: int, world: int) -> range:
    return range(rank, n_frames, world)


def allreduce_sum_(buf: torch.Tensor, group=None) -> torch.Tensor:
    """In-place sum over ranks of the gradient SoA (one collective for all 59 planes)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return buf


def replicas_in_sync(params: torch.Tensor, group=None) -> bool:
    """True when every rank holds bit-identical parameters (cheap checksum exchange)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return True
    chk = torch.stack([params.double().sum(), params.double().abs().sum()])
    lo, hi = chk.clone(), chk.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    return bool(torch.equal(lo, hi))

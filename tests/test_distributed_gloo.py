"""CPU, world_size 2 and 8 over gloo: the N>1 path of the engine -- views shard disjointly, the ONE
all-reduce over the gradient SoA gives the cross-rank sum (also on the low-plane slice of the compact
exchange, whose dL/dcolour planes are all-gathered in rank order), replicas that apply the same update stay
bit-identical, and frames shard without overlap."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from omfs_4d_video_gen_amd.engine.distributed import allgather_into_, allreduce_sum_, frames_of_rank, replicas_in_sync, view_index
    n_views, n_pad = 16, 512
    views = [view_index(step, rank, world, n_views) for step in range(8)]
    g = torch.Generator().manual_seed(100 + rank)
    params = torch.zeros(59, n_pad)
    for step in range(3):
        grads = torch.randn(59, n_pad, generator=g)
        local = grads.clone()
        allreduce_sum_(grads)
        gathered = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        assert torch.allclose(grads, torch.stack(gathered).sum(0), rtol=0, atol=1e-5)      # the ring adds in its own order
        assert replicas_in_sync(grads)
        params -= 0.01 * grads / world              # same update on every rank
        assert replicas_in_sync(params)
    # compact exchange: low-plane slice all-reduced in place, dL/dcolour planes gathered in rank order
    grads = torch.randn(59, n_pad, generator=g)
    local = grads.clone()
    allreduce_sum_(grads[:14])
    both = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(both, local)
    assert torch.allclose(grads[:14], torch.stack(both).sum(0)[:14], rtol=0, atol=1e-5) and torch.equal(grads[14:], local[14:])
    drgb = torch.full((3, n_pad), float(rank + 1))
    out = torch.zeros(world, 3, n_pad)
    allgather_into_(out, drgb)
    assert all(torch.equal(out[r], torch.full((3, n_pad), float(r + 1))) for r in range(world))
    # sharded exchange: reduce-scatter of the flat buffer (rank r owns elements [r*S, (r+1)*S)), update of the own shard,
    # all-gather of the updated shards
    from omfs_4d_video_gen_amd.engine.distributed import allgather_shards_, reduce_scatter_sum_
    grads = torch.randn(59, n_pad, generator=g)
    both = [torch.zeros_like(grads) for _ in range(world)]
    dist.all_gather(both, grads.clone())
    assert grads.numel() % (4 * world) == 0                       # 59 * n_pad / W is a whole number of float4 (n_pad = 512: W <= 128)
    S = grads.numel() // world
    shard = torch.empty(S)
    reduce_scatter_sum_(shard, grads.view(-1), None)
    total = torch.stack(both).sum(0)
    assert torch.allclose(shard, total.view(-1)[rank * S:(rank + 1) * S], rtol=0, atol=1e-5)
    p2 = params.clone().view(-1)
    p2[rank * S:(rank + 1) * S] -= 0.01 * shard / world          # only the own shard is updated ...
    allgather_shards_(p2, None)                                    # ... and every rank receives all of them
    assert replicas_in_sync(p2) and torch.allclose(p2, (params - 0.01 * total / world).view(-1), rtol=0, atol=1e-6)
    # the view set of one step as the compact exchange hands it to omfs_sh_rest_grads: one entry per rank, all distinct
    from omfs_4d_video_gen_amd import _lib as L
    vs = L.ViewSetC()
    vs.n_views = world
    for w in range(world):
        vs.view[w] = view_index(3, w, world, n_views)
    assert world <= len(vs.view) and len({vs.view[w] for w in range(world)}) == world
    # the shuffled schedule: every rank walks ITS views (index = rank mod W) without replacement, epoch by epoch
    from omfs_4d_video_gen_amd.engine.distributed import views_of_rank
    own = list(views_of_rank(n_views, rank, world))
    for epoch in range(3):
        seen = [view_index(epoch * len(own) + k, rank, world, n_views, seed=7) for k in range(len(own))]
        assert sorted(seen) == own
    bad = params + (rank * 1e-3)
    assert not replicas_in_sync(bad)
    q.put((rank, views, list(frames_of_rank(11, rank, world)), params.double().sum().item()))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])      # 8 = the node the scaling run uses (rehearsed here on CPU tensors only)
def test_data_parallel_plumbing(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    views = [r[1] for r in res]
    for step in range(8):        # the views of one step are pairwise distinct across the ranks ...
        assert len({v[step] for v in views}) == world
    # ... and 16 views are covered exactly once every 16 / W steps
    assert sorted(sum((v[:16 // world] for v in views), [])) == list(range(16))
    frames = [r[2] for r in res]
    assert sorted(sum(frames, [])) == list(range(11)) and sum(len(f) for f in frames) == 11
    assert len({r[3] for r in res}) == 1


def test_bench_gpus_n_starts_n_ranks_itself():
    """`python bench.py --gpus 2` without a launcher must start 2 ranks as children (torch.distributed.run on 127.0.0.1) and
    exit with their code.  No GPU here: every rank reports that and exits 2 -- but both must have started."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    argv = bench.launcher_argv(4, 29511, ["--gpus", "4", "--steps", "3"])
    assert argv[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in argv and "--nnodes=1" in argv
    assert argv[argv.index("--master-addr") + 1] == "127.0.0.1" and argv[-4:] == ["--gpus", "4", "--steps", "3"]
    assert argv[argv.index("--master-port") + 2].endswith("bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert "rank 0/2 started" in r.stderr and "rank 1/2 started" in r.stderr, r.stderr[-2000:]
    import torch
    if not torch.cuda.is_available():
        assert r.returncode != 0          # the children's failure is the parent's exit code
    # a launcher that started another number of ranks than --gpus says is an error, not a mislabelled line
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4"], capture_output=True, text=True,
                       env={**env, "WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"}, timeout=300)
    assert r.returncode == 2 and "WORLD_SIZE=2" in r.stdout

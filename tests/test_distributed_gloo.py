"""CPU, world_size 2 over gloo: the N>1 path of the engine -- views shard disjointly, the ONE
all-reduce over the gradient SoA gives the cross-rank sum (also on the 14-plane slice of the compact
exchange, whose dL/dcolour planes are all-gathered in rank order), replicas that apply the same update stay
bit-identical, and frames shard without overlap."""
import os
import socket

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
        assert torch.equal(grads, gathered[0] + gathered[1])
        params -= 0.01 * grads / world              # same update on every rank
        assert replicas_in_sync(params)
    # compact exchange: 14-plane slice all-reduced in place, dL/dcolour planes gathered in rank order
    grads = torch.randn(59, n_pad, generator=g)
    local = grads.clone()
    allreduce_sum_(grads[:14])
    both = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(both, local)
    assert torch.equal(grads[:14], (both[0] + both[1])[:14]) and torch.equal(grads[14:], local[14:])
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
    S = grads.numel() // world
    shard = torch.empty(S)
    reduce_scatter_sum_(shard, grads.view(-1), None)
    assert torch.equal(shard, (both[0] + both[1]).view(-1)[rank * S:(rank + 1) * S])
    p2 = params.clone().view(-1)
    p2[rank * S:(rank + 1) * S] -= 0.01 * shard / world          # only the own shard is updated ...
    allgather_shards_(p2, None)                                    # ... and every rank receives all of them
    assert replicas_in_sync(p2) and torch.equal(p2, (params - 0.01 * (both[0] + both[1]) / world).view(-1))
    bad = params + (rank * 1e-3)
    assert not replicas_in_sync(bad)
    q.put((rank, views, list(frames_of_rank(11, rank, world)), params.double().sum().item()))
    dist.destroy_process_group()


def test_two_rank_data_parallel_plumbing():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, v0, f0, s0), (_, v1, f1, s1) = res
    assert all(a != b for a, b in zip(v0, v1)) and sorted(v0 + v1) == sorted(list(range(16)))
    assert sorted(f0 + f1) == list(range(11)) and not set(f0) & set(f1)
    assert s0 == s1


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

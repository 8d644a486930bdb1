"""GPU: the data-parallel training step.  Two ranks (gloo, both on the one card of the test box) train
for a few steps with sharded views and the gradient all-reduce; they must stay bit-identical AND equal
a single process that sums the two views' gradients itself (to fp32 accumulation noise: the backward
pass adds with float atomics, whose order differs between runs)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
N, W, H, STEPS = 4000, 96, 64, 4


def _scene():
    from omfs_4d_video_gen_amd.engine import synthetic
    from omfs_4d_video_gen_amd.engine.flame_rig import FlameRig
    from omfs_4d_video_gen_amd.engine.trainer import View
    rig = synthetic.make_rig(0)
    seq = synthetic.make_flame_sequence(4, 0)
    g = synthetic.make_gaussians(N, rig.faces.shape[0], 0)
    gen = torch.Generator().manual_seed(5)
    views = [View(synthetic.make_camera(W, H, yaw=0.2 * i - 0.3), i, target=torch.rand(3, H, W, generator=gen).cuda()) for i in range(4)]
    return FlameRig.from_synthetic(rig), seq, g, views


def _worker(rank, world, port, q, exchange="compact", finetune=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      OMFS_DP_EXCHANGE=exchange)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from omfs_4d_video_gen_amd.engine.distributed import replicas_in_sync
    from omfs_4d_video_gen_amd.engine.trainer import Trainer
    rig, seq, g, views = _scene()
    tr = Trainer(rig, seq, g, views, W, H, start_sh_degree=3, rank=rank, world_size=world, process_group=dist.group.WORLD,
                 finetune_flame=finetune, flame_lr={"translation": 1e-4, "pose": 1e-4})
    assert tr.compact_dp == (exchange == "compact") and tr.sharded_dp == (exchange == "sharded")
    used = []
    for _ in range(STEPS):
        used.append(next(i for i, v in enumerate(tr.views) if v is tr.view_for_step(tr.step_idx)))
        tr.step()
    torch.cuda.synchronize()
    ok = replicas_in_sync(tr.model.params)
    if finetune:
        ok = ok and replicas_in_sync(tr.flame_ft.translation) and replicas_in_sync(tr.flame_ft.pose) and replicas_in_sync(tr.flame_ft.expr)
        moved = float((tr.flame_ft.translation.cpu() - torch.from_numpy(np.asarray(seq["translation"], np.float32).reshape(-1, 3))).abs().max())
        ok = ok and moved > 0
    q.put((rank, ok, used, tr.model.params.cpu().numpy()))
    dist.destroy_process_group()


def _run_two_ranks(exchange, finetune=False):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, exchange, finetune)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=60) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return res


def test_two_ranks_with_flame_finetuning_stay_in_sync():
    (_, ok0, _, p0), (_, ok1, _, p1) = _run_two_ranks("compact", finetune=True)
    assert ok0 and ok1 and np.array_equal(p0, p1)


@pytest.mark.parametrize("exchange", ["compact", "full", "sharded"])
def test_two_ranks_match_single_process_gradient_sum(exchange):
    res = _run_two_ranks(exchange)
    procs = []
    (_, ok0, used0, p0), (_, ok1, used1, p1) = res
    assert ok0 and ok1 and np.array_equal(p0, p1)
    assert used0 == [0, 2, 0, 2] and used1 == [1, 3, 1, 3]

    # single process: same two views per step, gradients summed locally, Adam with grad_scale 1/2
    from omfs_4d_video_gen_amd.engine.trainer import Trainer
    rig, seq, g, views = _scene()
    tr = Trainer(rig, seq, g, views, W, H, start_sh_degree=3)
    tr.use_graph = False                                  # the optimiser is swapped out below: eager iterations only
    total = torch.zeros_like(tr.grads)
    real_step = tr.opt.step
    for s in range(STEPS):
        total.zero_()
        for r in range(2):
            tr.step_idx = s
            tr.views_backup = tr.views
            v = views[(s * 2 + r) % 4]
            tr.views = [v]
            tr.opt.step = lambda grads, scale=1.0: None          # run everything but the optimiser
            tr.step()
            total += tr.grads
            tr.views = tr.views_backup
        tr.opt.step = real_step
        tr.step_idx = s + 1
        tr.opt.step(total, 0.5)
    torch.cuda.synchronize()
    # float atomics in the backward pass order their sums differently from run to run, so the two
    # executions agree to fp32 accumulation noise, not bitwise (the two RANKS are bitwise equal: see above)
    got = tr.model.params.cpu().numpy()
    assert np.allclose(got, p0, rtol=2e-4, atol=2e-6), np.abs(got - p0).max()


def _worker_rccl(port, q, exchange):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", OMFS_DP_EXCHANGE=exchange,
                      OMFS_DP_FORCE="1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from omfs_4d_video_gen_amd.engine.trainer import Trainer
    rig, seq, g, views = _scene()
    tr = Trainer(rig, seq, g, views, W, H, start_sh_degree=3, rank=0, world_size=1, process_group=dist.group.WORLD)
    assert tr.dp and tr.compact_dp == (exchange == "compact") and tr.sharded_dp == (exchange == "sharded")
    for _ in range(STEPS):
        tr.step()
    torch.cuda.synchronize()
    tr.rast.check_status()
    q.put(tr.model.params.cpu().numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("exchange", ["compact", "full", "sharded"])
def test_exchange_path_over_rccl_with_one_rank_equals_the_plain_step(exchange):
    """The collectives of the data-parallel step issued on the real backend ("nccl" = RCCL; one rank is all a one-GPU
    box has): asynchronous all-gather under project_bwd, asynchronous all-reduce of the 14 planes under the SH update.
    With one rank the sums are the rank's own gradients, so the trajectory is the single-GPU one."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    p = ctx.Process(target=_worker_rccl, args=(port, q, exchange))
    p.start()
    got = q.get(timeout=120)
    p.join(60)
    assert p.exitcode == 0
    from omfs_4d_video_gen_amd.engine.trainer import Trainer
    rig, seq, g, views = _scene()
    tr = Trainer(rig, seq, g, views, W, H, start_sh_degree=3)
    assert not tr.dp
    for _ in range(STEPS):
        tr.step()
    torch.cuda.synchronize()
    want = tr.model.params.cpu().numpy()
    assert np.allclose(got, want, rtol=2e-4, atol=2e-6), np.abs(got - want).max()


def test_three_exchange_modes_give_the_same_replicas():
    """compact (14-plane all-reduce + dL/dcolour all-gather), full (one all-reduce) and sharded (reduce-scatter, Adam on 1/W of
    the elements, all-gather of the parameters): the ranks of each run are bit-identical, and the three runs end in the same
    parameters up to the order in which the float atomics of the backward pass and the collectives add."""
    ends = {}
    for mode in ("compact", "full", "sharded"):
        (_, ok0, _, p0), (_, ok1, _, p1) = _run_two_ranks(mode)
        assert ok0 and ok1 and np.array_equal(p0, p1), mode
        ends[mode] = p0
    for mode in ("full", "sharded"):
        assert np.allclose(ends[mode], ends["compact"], rtol=2e-4, atol=2e-6), (mode, np.abs(ends[mode] - ends["compact"]).max())

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


def _worker(rank, world, port, q, exchange="compact", finetune=False, resume_after=0):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      OMFS_DP_EXCHANGE=exchange)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from omfs_4d_video_gen_amd.engine.distributed import replicas_in_sync
    from omfs_4d_video_gen_amd.engine.trainer import Trainer
    rig, seq, g, views = _scene()

    def make():
        return Trainer(rig, seq, g, views, W, H, start_sh_degree=3, rank=rank, world_size=world, process_group=dist.group.WORLD,
                       finetune_flame=finetune, flame_lr={"translation": 1e-4, "pose": 1e-4})
    tr = make()
    assert tr.compact_dp == (exchange == "compact") and tr.sharded_dp == (exchange == "sharded")
    used = []
    for _ in range(STEPS):
        used.append(next(i for i, v in enumerate(tr.views) if v is tr.view_for_step(tr.step_idx)))
        tr.step()
    torch.cuda.synchronize()
    extra = None
    if resume_after:
        # what engine/train.py does at a checkpoint: the moments are made whole on every rank (a collective in the sharded
        # exchange), rank 0's copy is what a checkpoint holds; a fresh trainer on every rank loads it and goes on
        tr.sync_optimizer_state()
        moments_whole = replicas_in_sync(tr.opt.m) and replicas_in_sync(tr.opt.v)
        state = [tr.model.params.clone(), tr.opt.m.clone(), tr.opt.v.clone(), tr.opt.step_count, tr.step_idx]
        dist.broadcast_object_list(obj := [[t.cpu() if torch.is_tensor(t) else t for t in state]], src=0)
        p_, m_, v_, n_, i_ = obj[0]
        tr2 = make()
        tr2.model.params.copy_(p_); tr2.opt.m.copy_(m_); tr2.opt.v.copy_(v_)
        tr2.opt.step_count, tr2.step_idx = n_, i_
        for t in (tr, tr2):
            for _ in range(resume_after):
                t.step()
        torch.cuda.synchronize()
        extra = (moments_whole, tr2.model.params.cpu().numpy(), replicas_in_sync(tr2.model.params))
    ok = replicas_in_sync(tr.model.params)
    if finetune:
        ok = ok and replicas_in_sync(tr.flame_ft.translation) and replicas_in_sync(tr.flame_ft.pose) and replicas_in_sync(tr.flame_ft.expr)
        moved = float((tr.flame_ft.translation.cpu() - torch.from_numpy(np.asarray(seq["translation"], np.float32).reshape(-1, 3))).abs().max())
        ok = ok and moved > 0
    q.put((rank, ok, used, tr.model.params.cpu().numpy(), extra))
    dist.destroy_process_group()


def _run_ranks(exchange, finetune=False, world=2, resume_after=0):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, exchange, finetune, resume_after)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return res


def _run_two_ranks(exchange, finetune=False):
    return [r[:4] for r in _run_ranks(exchange, finetune)]


def test_two_ranks_with_flame_finetuning_stay_in_sync():
    (_, ok0, _, p0), (_, ok1, _, p1) = _run_two_ranks("compact", finetune=True)
    assert ok0 and ok1 and np.array_equal(p0, p1)


def _single_process_sum(world):
    """One process: the `world` views of every step, gradients summed locally, Adam with grad_scale 1 / world."""
    from omfs_4d_video_gen_amd.engine.trainer import Trainer
    rig, seq, g, views = _scene()
    tr = Trainer(rig, seq, g, views, W, H, start_sh_degree=3)
    tr.use_graph = False                                  # the optimiser is swapped out below: eager iterations only
    tr.sh_adam = False                                    # ... and it takes every plane from the gradient buffer this function sums
    total = torch.zeros_like(tr.grads)
    real_step = tr.opt.step
    for s in range(STEPS):
        total.zero_()
        for r in range(world):
            tr.step_idx = s
            tr.views_backup = tr.views
            v = views[(s * world + r) % 4]
            tr.views = [v]
            tr.opt.step = lambda grads, scale=1.0: None          # run everything but the optimiser
            tr.step()
            total += tr.grads
            tr.views = tr.views_backup
        tr.opt.step = real_step
        tr.step_idx = s + 1
        tr.opt.step(total, 1.0 / world)
    torch.cuda.synchronize()
    return tr.model.params.cpu().numpy()


@pytest.mark.parametrize("exchange", ["compact", "full", "sharded"])
def test_two_ranks_match_single_process_gradient_sum(exchange):
    res = _run_two_ranks(exchange)
    (_, ok0, used0, p0), (_, ok1, used1, p1) = res
    assert ok0 and ok1 and np.array_equal(p0, p1)
    assert used0 == [0, 2, 0, 2] and used1 == [1, 3, 1, 3]
    # float atomics in the backward pass order their sums differently from run to run, so the two
    # executions agree to fp32 accumulation noise, not bitwise (the two RANKS are bitwise equal: see above)
    got = _single_process_sum(2)
    assert np.allclose(got, p0, rtol=2e-4, atol=2e-6), np.abs(got - p0).max()


@pytest.mark.parametrize("exchange", ["full", "sharded"])
def test_two_ranks_equal_the_single_process_sum_bit_for_bit_when_deterministic(exchange, monkeypatch):
    """OMFS_DETERMINISTIC=1 makes every gradient a function of the parameters alone (no float-atomic order), so two ranks that
    all-reduce their views' gradients end on EXACTLY the parameters of one process that adds the same two gradient buffers itself
    (a + b is the one floating-point sum both form; the compact exchange rebuilds its SH planes with fused multiply-adds over the
    views and is only close).  The statistical form of this comparison, in the default mode, is the test above."""
    monkeypatch.setenv("OMFS_DETERMINISTIC", "1")
    (_, ok0, _, p0), (_, ok1, _, p1) = _run_two_ranks(exchange)
    assert ok0 and ok1 and np.array_equal(p0, p1)
    got = _single_process_sum(2)
    assert np.array_equal(got, p0), float(np.abs(got - p0).max())


@pytest.mark.parametrize("exchange", ["compact", "full", "sharded"])
def test_four_ranks_on_one_card_match_the_four_view_gradient_sum(exchange):
    """The widest rehearsal one card allows (the box admits 6 GPU processes; world size 8 itself is covered on CPU tensors in
    tests/test_distributed_gloo.py and has never run on hardware): 4 gloo ranks, every step consumes all 4 views."""
    res = _run_ranks(exchange, world=4)
    assert all(r[1] for r in res) and all(np.array_equal(res[0][3], r[3]) for r in res[1:])
    assert [r[2] for r in res] == [[k] * STEPS for k in range(4)]
    got = _single_process_sum(4)
    assert np.allclose(got, res[0][3], rtol=2e-4, atol=2e-6), np.abs(got - res[0][3]).max()


def test_sharded_exchange_resumes_from_whole_moments_like_a_continuous_run():
    """A rank of the sharded exchange maintains Adam's moments for its own 1/W of the elements only: Trainer.sync_optimizer_state()
    makes them whole (engine/train.py calls it before a checkpoint, engine/densify.py before a compaction).  A run resumed from
    rank 0's synced state must go on exactly like the run that was never interrupted."""
    res = _run_ranks("sharded", resume_after=3)
    for _, ok, _, p_cont, (whole, p_resumed, in_sync) in res:
        assert ok and whole and in_sync
        assert np.allclose(p_resumed, p_cont, rtol=2e-4, atol=2e-6), np.abs(p_resumed - p_cont).max()


def _worker_rccl(port, q, exchange):
    impl = "torch"
    if exchange.endswith("-abi"):         # the exchange issued by the library's own communicator (omfs_rccl_allreduce_grads / _allgather)
        exchange, impl = exchange[:-4], "abi"
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", OMFS_DP_EXCHANGE=exchange,
                      OMFS_DP_FORCE="1", OMFS_DP_IMPL=impl)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from omfs_4d_video_gen_amd.engine.trainer import Trainer
    rig, seq, g, views = _scene()
    tr = Trainer(rig, seq, g, views, W, H, start_sh_degree=3, rank=0, world_size=1, process_group=dist.group.WORLD)
    assert tr.dp and tr.compact_dp == (exchange == "compact") and tr.sharded_dp == (exchange == "sharded")
    assert (tr._abi_comm is not None) == (impl == "abi")
    for _ in range(STEPS):
        tr.step()
    torch.cuda.synchronize()
    tr.rast.check_status()
    q.put(tr.model.params.cpu().numpy())
    if tr._abi_comm is not None:
        tr._abi_comm.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("exchange", ["compact", "full", "sharded", "full-abi", "compact-abi"])
def test_exchange_path_over_rccl_with_one_rank_equals_the_plain_step(exchange):
    """The collectives of the data-parallel step issued on the real backend ("nccl" = RCCL; one rank is all a one-GPU
    box has): asynchronous all-gather under project_bwd, asynchronous all-reduce of the 11 geometry / opacity planes under the SH update.
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


def _worker_abi_comm(q):
    torch.cuda.set_device(0)
    from omfs_4d_video_gen_amd.engine.distributed import AbiComm
    comm = AbiComm(0, 1)
    x = torch.arange(4096, dtype=torch.float32, device="cuda") * 0.25
    ref = x.clone()
    comm.allreduce_(x)                                      # sum over one rank
    out = torch.zeros_like(ref)
    comm.allgather_(out, ref)
    shard = torch.zeros_like(ref)
    comm.reduce_scatter_(shard, ref)
    torch.cuda.synchronize()
    ok = bool(torch.equal(x, ref) and torch.equal(out, ref) and torch.equal(shard, ref))
    comm.close()
    q.put(ok)


def test_c_abi_communicator_collectives_with_one_rank():
    """omfs_comm_* / omfs_rccl_* (RCCL bound by dlopen inside libomfs_splat.so, no torch.distributed anywhere): with one rank
    every collective is the identity.  In a child process: RCCL keeps threads and device state of its own."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker_abi_comm, args=(q,))
    p.start()
    assert q.get(timeout=180)
    p.join(60)
    assert p.exitcode == 0


def test_three_exchange_modes_give_the_same_replicas():
    """compact (11-plane all-reduce + dL/dcolour all-gather), full (one all-reduce) and sharded (reduce-scatter, Adam on 1/W of
    the elements, all-gather of the parameters): the ranks of each run are bit-identical, and the three runs end in the same
    parameters up to the order in which the float atomics of the backward pass and the collectives add."""
    ends = {}
    for mode in ("compact", "full", "sharded"):
        (_, ok0, _, p0), (_, ok1, _, p1) = _run_two_ranks(mode)
        assert ok0 and ok1 and np.array_equal(p0, p1), mode
        ends[mode] = p0
    for mode in ("full", "sharded"):
        assert np.allclose(ends[mode], ends["compact"], rtol=2e-4, atol=2e-6), (mode, np.abs(ends[mode] - ends["compact"]).max())


def test_bench_two_ranks_end_to_end_on_one_card():
    """`python bench.py --gpus 2 ...` end to end: the parent starts two ranks, both share the box's one card
    (OMFS_DIST_BACKEND=gloo: functional rehearsal, no RCCL link involved), rank 0 prints ONE JSON line that says what ran."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(OMFS_DIST_BACKEND="gloo", OMFS_DP_EXCHANGE="compact")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--no_cpu_baseline",
                        "--n_gaussians", "60000", "--width", "640", "--height", "360", "--render_frames", "8", "--profile_steps", "4"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 5 and out["warmup"] == 2 and out["scaling"] == "weak"
    assert out["replicas_in_sync"] is True
    assert out["config"]["parallelism"].startswith("dp2") and "all-gather" in out["config"]["parallelism"]     # names the exchange
    assert out["value"] > 0 and abs(out["value"] - 2 * 5 / (out["ms_per_step"] * 5e-3)) < 1e-2 * out["value"]
    assert "allreduce" in out["stages_ms"] and out["roofline"]["kernel"] != "allreduce"
    assert out["aux"]["render_surgery_fps"] > 0

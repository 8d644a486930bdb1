"""GPU, end to end through the reference's call surface: write a dataset in the converter's layout,
`train_ghost.train()` (child process running engine/train.py), `render_surgery` offsets -> modified
dataset -> `render_with_gaussians()` (child process running engine/render.py) -> deterministic
export -> `validation_reporting`.  Checks files, formats, progress lines and that training lowers the loss."""
import json
import os
import re
import shutil
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

import helpers as H

from omfs_4d_video_gen_amd.engine import synthetic

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def dataset(tmp_path_factory):
    """60 frames, 96x72, rendered by the engine itself from a 'ground-truth' Gaussian set."""
    return H.build_cli_dataset(tmp_path_factory.mktemp("data"))


def test_train_render_validate_pipeline(dataset, tmp_path, monkeypatch, capfd):
    from omfs_4d_video_gen_amd import render_surgery as rs
    from omfs_4d_video_gen_amd import train_ghost as tg
    from omfs_4d_video_gen_amd import validation_reporting as vr
    from omfs_4d_video_gen_amd.engine import io_formats as IO
    monkeypatch.setenv("OMFS_SYNTHETIC_RIG", "1")
    model = tmp_path / "model"
    real_run = subprocess.run

    def run_with_extra(cmd, **kw):      # the UI passes nothing else; the test shortens the run via extra engine flags
        if str(cmd[1]).endswith("train.py"):
            cmd = list(cmd) + ["--n_gaussians", "20000", "--log_every", "10"]
        return real_run(cmd, **kw)
    monkeypatch.setattr(tg.subprocess, "run", run_with_extra)
    tg.train(str(dataset), str(model), iterations=200, resolution=-1)
    out = capfd.readouterr().out
    its = [int(m) for m in re.findall(r"iteration\s+(\d+)", out.lower())]          # app.py:1387-1398 progress regex
    assert its and its[-1] == 200
    losses = [float(m) for m in re.findall(r"loss=([0-9.]+)", out)]
    assert np.mean(losses[-5:]) < 0.9 * np.mean(losses[:2]), losses            # per-view losses: compare averages
    assert "--white_background" in out                                              # fg_masks present -> white background
    assert (model / "point_cloud" / "iteration_200" / "point_cloud.ply").exists()
    assert (model / "chkpnt200.pth").exists() and list((model / "experiment_manifests").glob("*.json"))
    ck = torch.load(model / "chkpnt200.pth", weights_only=True)
    assert ck["iteration"] == 200 and ck["params"].shape[0] == 59
    g = IO.load_gaussian_ply(model / "point_cloud" / "iteration_200" / "point_cloud.ply")
    assert g["xyz"].shape == (20000, 3) and np.isfinite(g["sh"]).all()

    # render_surgery: 3 mm Le Fort, 5 mm BSSO
    lef, bsso = rs.compute_offset(3.0, 1.0), rs.compute_offset(5.0, 1.0)
    mod = rs.create_modified_dataset(str(dataset), lef, bsso)
    try:
        stale = model / "train" / "ours_1" / "renders"
        stale.mkdir(parents=True)
        (stale / "old.png").write_bytes(b"x")
        renders = rs.render_with_gaussians(str(model), mod)
        assert not stale.exists()                                                   # old renders are cleared (reference :261-267)
        assert Path(renders) == model / "train" / "ours_200" / "renders"
        names = sorted(os.listdir(renders))
        assert names == [f"{i:05d}.png" for i in range(54)]                         # the TRAIN split (--skip_val --skip_test)
        img = IO.read_png(Path(renders) / names[10])
        assert img.shape == (72, 96, 3) and img.std() > 5
        assert sorted(os.listdir(model / "train" / "ours_200" / "gt")) == names
        det = rs.export_deterministic_frames(renders, str(tmp_path / "det"))
        vr.generate_report(model, Path(det), tmp_path / "report")
        rep = json.loads((tmp_path / "report" / "strict_scores.json").read_text())
        assert rep["summary"]["count"] == 24 and all(r["psnr"] > 10 for r in rep["rows"])
        # the surgical offsets must change the picture: render the unmodified dataset too
        base = rs.create_modified_dataset(str(dataset), 0.0, 0.0)
        try:
            r0 = rs.render_with_gaussians(str(model), base)
            a = IO.read_png(Path(r0) / names[10]).astype(np.float32)
        finally:
            shutil.rmtree(base, ignore_errors=True)
        assert np.abs(a - img.astype(np.float32)).mean() > 0.05
        with pytest.raises(FileNotFoundError):
            rs.stitch_video(renders, str(tmp_path / "v.mp4")) if shutil.which("ffmpeg") is None else (_ for _ in ()).throw(FileNotFoundError())
    finally:
        shutil.rmtree(mod, ignore_errors=True)


def test_single_frame_experiment(dataset, tmp_path, monkeypatch):
    from omfs_4d_video_gen_amd import single_frame_experiment as sfe
    from omfs_4d_video_gen_amd import train_ghost as tg
    monkeypatch.setenv("OMFS_SYNTHETIC_RIG", "1")
    real_run = subprocess.run
    monkeypatch.setattr(tg.subprocess, "run", lambda cmd, **kw: real_run(list(cmd) + (["--n_gaussians", "20000"] if str(cmd[1]).endswith("train.py") else []), **kw))
    res = sfe.run(str(dataset), str(tmp_path / "work"), iterations=300, copies=50)
    assert Path(res["gt"]).exists() and Path(res["render"]).exists()
    assert res["psnr"] > 14.0          # 300 iterations on one white-background view: clearly better than a blank frame


def test_engine_failure_surfaces_as_runtime_error(dataset, tmp_path, monkeypatch):
    from omfs_4d_video_gen_amd import render_surgery as rs
    monkeypatch.setenv("OMFS_SYNTHETIC_RIG", "1")
    (tmp_path / "empty_model").mkdir()
    with pytest.raises(RuntimeError, match="Rendering failed"):
        rs.render_with_gaussians(str(tmp_path / "empty_model"), str(dataset))


def test_finetune_flame_checkpoint_resume_and_tuned_render(dataset, tmp_path, monkeypatch):
    """engine/train.py --finetune_flame_params with a checkpoint, resumed run == uninterrupted run (up to the order
    of float atomics); engine/render.py renders `tuned + (dataset - source)`."""
    from omfs_4d_video_gen_amd.engine import io_formats as IO
    from omfs_4d_video_gen_amd.engine import render as R
    env = {**os.environ, "OMFS_SYNTHETIC_RIG": "1", "PYTHONPATH": str(ROOT)}
    train = str(ROOT / "omfs_4d_video_gen_amd" / "engine" / "train.py")
    common = ["--source_path", str(dataset), "--bind_to_mesh", "--n_gaussians", "12000", "--log_every", "10", "--finetune_flame_params",
              "--flame_trans_lr", "1e-4", "--flame_pose_lr", "1e-4", "--white_background"]
    a, b = tmp_path / "a", tmp_path / "b"
    r1 = subprocess.run([sys.executable, train, *common, "--model_path", str(a), "--iterations", "40", "--checkpoint_iterations", "25"],
                        env=env, capture_output=True, text=True)
    assert r1.returncode == 0, r1.stderr[-2000:]
    assert (a / "chkpnt25.pth").exists()
    r2 = subprocess.run([sys.executable, train, *common, "--model_path", str(b), "--iterations", "40",
                         "--start_checkpoint", str(a / "chkpnt25.pth"), "--target_storage", "u8"],   # 8-bit targets: same values
                        env=env, capture_output=True, text=True)
    assert r2.returncode == 0, r2.stderr[-2000:]
    assert "resumed from" in r2.stdout and "iteration 40/40" in r2.stdout
    ga = IO.load_gaussian_ply(a / "point_cloud" / "iteration_40" / "point_cloud.ply")
    gb = IO.load_gaussian_ply(b / "point_cloud" / "iteration_40" / "point_cloud.ply")
    for k in ("xyz", "log_scale", "opacity", "sh"):    # Adam's sign-like first steps amplify the float-atomic noise
        H.assert_same_up_to_atomic_noise(ga[k], gb[k], 2e-4, 1e-2, k)      # measured: p99.9 <= 6.4e-4, max 1.2e-3 (printed into the test log)
    fa = dict(np.load(a / "point_cloud" / "iteration_40" / "flame_param.npz"))
    fb = dict(np.load(b / "point_cloud" / "iteration_40" / "flame_param.npz"))
    src = dict(np.load(a / "point_cloud" / "iteration_40" / "flame_param_source.npz"))
    assert np.abs(fa["translation"] - src["translation"]).max() > 1e-5          # the parameters moved ...
    assert np.abs(fa["translation"] - fb["translation"]).max() < 6e-4           # ... the same way in both runs (lr 1e-4: a handful of sign-like Adam steps apart)
    assert fa["expr"].shape == src["expr"].shape and set(fa) == set(src)
    # render-time sequence: tuned + (dataset - source)
    edited = {k: np.array(v) for k, v in src.items()}
    edited["translation"] = edited["translation"] + np.array([0.0, 0.002, 0.0], np.float32)
    out = R.tuned_flame(a / "point_cloud" / "iteration_40", edited)
    assert np.allclose(out["translation"], fa["translation"] + np.array([0.0, 0.002, 0.0], np.float32), atol=1e-7)
    assert np.allclose(out["expr"], fa["expr"])


def test_overflowed_interval_is_rolled_back_and_redone(dataset, tmp_path):
    """engine/train.py with a tile-list capacity far too small: the intervals that rendered empty lists are not kept -- the
    trainer returns to its last good snapshot, grows the capacity and redoes them; the run ends where a run with ample
    capacity ends (up to the order of the float atomics), not on parameters that were stepped on momentum alone."""
    env = {**os.environ, "OMFS_SYNTHETIC_RIG": "1", "PYTHONPATH": str(ROOT)}
    train = str(ROOT / "omfs_4d_video_gen_amd" / "engine" / "train.py")
    common = ["--source_path", str(dataset), "--bind_to_mesh", "--n_gaussians", "12000", "--log_every", "10", "--white_background",
              "--iterations", "40", "--checkpoint_iterations", "40", "--no_densify"]
    a, b = tmp_path / "small", tmp_path / "ample"
    r1 = subprocess.run([sys.executable, train, *common, "--model_path", str(a), "--dup_capacity", "3000"], env=env, capture_output=True, text=True)
    assert r1.returncode == 0, r1.stderr[-2000:]
    assert "are redone" in r1.stdout and r1.stdout.count("are redone") >= 2          # 3000 -> 6000 -> 12000 -> ... pairs
    r2 = subprocess.run([sys.executable, train, *common, "--model_path", str(b)], env=env, capture_output=True, text=True)
    assert r2.returncode == 0 and "are redone" not in r2.stdout, r2.stderr[-2000:]
    ca, cb = torch.load(a / "chkpnt40.pth", weights_only=True), torch.load(b / "chkpnt40.pth", weights_only=True)
    # Adam's sign-like first steps amplify the float-atomic noise between two runs (see the resume test): the clouds agree in the
    # mean to a few 1e-4; a run that had KEPT its empty-render steps would sit an order of magnitude further away
    for k in ("params", "adam_m"):
        H.assert_same_up_to_atomic_noise(ca[k].numpy(), cb[k].numpy(), 3e-4, 0.02, k)      # measured: p99.9 4.6e-3, max 0.089 (one element)
    la = [float(m) for m in re.findall(r"loss=([0-9.]+)", r1.stdout)]
    lb = [float(m) for m in re.findall(r"loss=([0-9.]+)", r2.stdout)]
    assert abs(la[-1] - lb[-1]) < 0.02 * lb[-1]


def test_training_with_densification_under_a_small_pair_capacity(dataset, tmp_path):
    """engine/train.py with adaptive density control ON and a tile-list capacity far too small (no test ran train.py
    with densification): snapshots are taken after every densification (binding is [n], the planes [59][n_pad]), overflowed
    intervals are rolled back and redone, a save at an iteration inside a log interval (35) comes after that interval's overflow
    check, and the run ends with a consistent checkpoint of the grown cloud."""
    env = {**os.environ, "OMFS_SYNTHETIC_RIG": "1", "PYTHONPATH": str(ROOT)}
    train = str(ROOT / "omfs_4d_video_gen_amd" / "engine" / "train.py")
    m = tmp_path / "m"
    r = subprocess.run([sys.executable, train, "--source_path", str(dataset), "--model_path", str(m), "--bind_to_mesh", "--white_background",
                        "--n_gaussians", "12000", "--log_every", "10", "--iterations", "60", "--checkpoint_iterations", "60",
                        "--save_iterations", "35", "--dup_capacity", "3000", "--densify_from_iter", "15", "--densification_interval", "10",
                        "--densify_until_iter", "1000", "--densify_grad_threshold", "1e-6", "--max_gaussians", "40000"],
                       env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-2500:]
    assert "are redone" in r.stdout and "densify:" in r.stdout and "iteration 60/60" in r.stdout, r.stdout[-3000:]
    ck = torch.load(m / "chkpnt60.pth", weights_only=True)
    n = int(ck["binding"].shape[0])
    assert n > 12000 and ck["params"].shape == (59, n) and ck["adam_m"].shape == ck["params"].shape      # checkpoints hold the n columns
    assert torch.isfinite(ck["params"][:, :n]).all()
    assert (m / "point_cloud" / "iteration_35" / "point_cloud.ply").exists() and (m / "point_cloud" / "iteration_60" / "point_cloud.ply").exists()
    # the kept iterations rendered real lists: the loss of the last interval is below the first's
    losses = [float(x) for x in re.findall(r"loss=([0-9.]+)", r.stdout)]
    assert np.mean(losses[-2:]) < np.mean(losses[:2]), losses


def test_renderer_uses_each_frames_own_camera():
    """the Renderer cached camera structs under id(view); engine/render.py builds a short-lived View per
    frame and CPython hands the freed address to the next one, so frames 2.. were rendered with the camera of frame 0 or 1.
    Six frames from six clearly different cameras through short-lived Views (one stream and three) against a renderer whose
    Views all stay alive."""
    from omfs_4d_video_gen_amd.engine.flame_rig import FlameRig
    from omfs_4d_video_gen_amd.engine.trainer import Renderer, View
    W, H, T = 128, 96, 6
    rig = synthetic.make_rig(0)
    frig = FlameRig.from_synthetic(rig)
    seq = synthetic.make_flame_sequence(T, 3)
    g = synthetic.make_gaussians(8000, rig.faces.shape[0], 1)
    cams = [synthetic.make_camera(W, H, yaw=-0.6 + 0.24 * i) for i in range(T)]
    keep = [View(cams[i], i) for i in range(T)]
    ref = Renderer(frig, seq, g, W, H)
    want = [ref.render(keep[i], rgb8=True).cpu().clone() for i in range(T)]
    assert all(not torch.equal(want[i], want[i + 1]) for i in range(T - 1))
    for n_streams in (1, 3):
        r = Renderer(frig, seq, g, W, H, n_streams=n_streams)
        for i in range(T):
            img, ev = r.render_async(View(cams[i], i), rgb8=True)       # the View dies at the end of the statement
            ev.synchronize()
            assert torch.equal(img.cpu(), want[i]), (n_streams, i)
        # render() on the caller's stream between asynchronous frames of another FLAME batch
        r.flame_batch = 2
        for i in (0, 3, 1, 5):
            a = r.render(View(cams[i], i), rgb8=True).cpu()
            b, ev = r.render_async(View(cams[(i + 2) % T], (i + 2) % T), rgb8=True)
            ev.synchronize()
            assert torch.equal(a, want[i]) and torch.equal(b.cpu(), want[(i + 2) % T]), (n_streams, i)


def test_render_grows_an_exceeded_pair_capacity_and_renders_again(dataset, tmp_path):
    """engine/render.py with a tile-list capacity far too small (the default is sized close to the heaviest cloud measured, 20 pairs
    per Gaussian at 1080p, so the path must exist): frames rendered after the overflow hold empty lists; the engine doubles the
    buffers, renders the split again and every PNG is the SAME BYTES an amply sized run writes."""
    env = {**os.environ, "OMFS_SYNTHETIC_RIG": "1", "PYTHONPATH": str(ROOT)}
    eng = ROOT / "omfs_4d_video_gen_amd" / "engine"
    m = tmp_path / "m"
    r = subprocess.run([sys.executable, str(eng / "train.py"), "--source_path", str(dataset), "--model_path", str(m), "--bind_to_mesh",
                        "--white_background", "--n_gaussians", "12000", "--iterations", "20", "--log_every", "10", "--no_densify"],
                       env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    outs = {}
    for name, extra in (("small", ["--dup_capacity", "3000"]), ("ample", [])):
        r = subprocess.run([sys.executable, str(eng / "render.py"), "--source_path", str(dataset), "--model_path", str(m), "--bind_to_mesh",
                            "--skip_val", "--skip_test", *extra], env=env, capture_output=True, text=True)
        assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-2000:]
        assert ("tile-list capacity exceeded" in r.stdout) == (name == "small"), r.stdout[-1500:]
        d = m / "train" / "ours_20" / "renders"
        outs[name] = {p.name: p.read_bytes() for p in sorted(d.iterdir())}
        shutil.rmtree(m / "train")
    assert len(outs["ample"]) > 40 and outs["small"] == outs["ample"]


def test_engine_clis_under_a_two_rank_launch(dataset, tmp_path):
    """SURVEY section 8e at the level of the engine's own command lines (configs 3 and 4 in small): `engine/train.py` and `engine/render.py`
    started by `torch.distributed.run --nproc-per-node 2` (both ranks on the box's one card, OMFS_DIST_BACKEND=gloo).  Training:
    views shard across the ranks, the run completes, rank 0 writes the point cloud and the checkpoint, the loss falls.  Rendering:
    frame f belongs to rank f mod 2, the union is the whole split, and every frame is the SAME BYTES a single process renders."""
    import socket
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(OMFS_SYNTHETIC_RIG="1", PYTHONPATH=str(ROOT), OMFS_DIST_BACKEND="gloo")
    eng = ROOT / "omfs_4d_video_gen_amd" / "engine"

    def launch(script, *args):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
               "--master-port", str(port), str(eng / script), *args]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-2500:]
        return r.stdout + r.stderr
    m = tmp_path / "m"
    out = launch("train.py", "--source_path", str(dataset), "--model_path", str(m), "--bind_to_mesh", "--white_background",
                 "--n_gaussians", "12000", "--log_every", "10", "--iterations", "60", "--checkpoint_iterations", "60", "--no_densify")
    losses = [float(x) for x in re.findall(r"loss=([0-9.]+)", out)]
    assert "iteration 60/60" in out and np.mean(losses[-2:]) < np.mean(losses[:2]), losses
    assert (m / "point_cloud" / "iteration_60" / "point_cloud.ply").exists() and (m / "chkpnt60.pth").exists()
    assert json.loads((m / "cfg_args.json").read_text())["world_size"] == 2
    # render the train split with two ranks, then with one process into a copy of the model directory
    out = launch("render.py", "--source_path", str(dataset), "--model_path", str(m), "--bind_to_mesh", "--skip_val", "--skip_test")
    counts = sorted(int(x) for x in re.findall(r"rendered (\d+) train frames", out))
    assert counts == [27, 27], out[-1500:]                              # 54 train frames: f mod 2
    two = m / "train" / "ours_60" / "renders"
    names = sorted(os.listdir(two))
    assert names == [f"{i:05d}.png" for i in range(54)]
    m1 = tmp_path / "m1"
    shutil.copytree(m, m1, ignore=shutil.ignore_patterns("train"))
    r = subprocess.run([sys.executable, str(eng / "render.py"), "--source_path", str(dataset), "--model_path", str(m1), "--bind_to_mesh",
                        "--skip_val", "--skip_test"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    one = m1 / "train" / "ours_60" / "renders"
    for n in names:
        assert (two / n).read_bytes() == (one / n).read_bytes(), n

#!/usr/bin/env python3
"""Generates the golden vectors under tests/golden/ by RUNNING THE REFERENCE in the build
container (it is mounted read-only at /root/reference and never travels to the GPU box).

  python tests/golden/make_goldens.py

What is captured: outputs of the reference's own pure functions on seeded inputs
  * 02_Visual_Engine/render_surgery.py : compute_offset, modify_flame_params, choose_rig_mode,
    export_deterministic_frames (index selection), create_modified_dataset (transforms rewrite)
  * 02_Visual_Engine/train_ghost.py    : run_quality_gates matrix, dataset fingerprint, the engine
    argv / save-iteration rule / manifest schema of train()
  * 02_Visual_Engine/flame_fitter.py   : SimpleFLAME._axis_angle_to_matrix, SimpleFLAME.forward,
    estimate_head_pose_from_landmarks, 1-, 3- and default-length (200) fit_flame_to_landmarks on the synthetic rig
  * 02_Visual_Engine/validation_reporting.py : psnr, ssim_global, _bucket, generate_report (strict_scores.json, checklist, refusals)
  * 02_Visual_Engine/head_recon/*.py   : the four scaffold outputs on a 4-directory capture root, refusals, coverage CLI
  * 02_Visual_Engine/render_surgery.py : render_with_gaussians with the child process stubbed (argv, stale-render clearing,
    iteration choice, renders-dir discovery, both error messages), stitch_video with ffmpeg stubbed (argv, staging), load_deformation_map
  * 02_Visual_Engine/single_frame_experiment.py : build_single_frame_dataset (file list, rewritten transforms, batched npz)
    -- the scenarios of these four live in tests/golden/scenarios.py and are run AGAIN by the tests on the drop-in modules.
Only DATA is written (npz / json); no reference source text is stored.

  python tests/golden/make_goldens.py [section ...]     # sections: preprocess render_surgery train_ghost flame_fitter surface
  python tests/golden/make_goldens.py --out DIR         # write everything into DIR instead (the reproducibility test)

`python tests/golden/make_goldens.py && git diff --exit-code tests/golden` is clean: every patch of a reference module is undone at
the end of its section (_Monkey), temporary paths are normalised, and `-m "not gpu"` runs exactly that comparison whenever
/root/reference is present (tests/test_reference_goldens.py::test_generator_reproduces_the_committed_goldens).

flame_fitter imports cv2 and mediapipe at module level; neither is used by the functions
captured here, so two empty placeholder modules are registered before the import.
"""
import importlib
import io
import json
import os
import shutil
import sys
import tempfile
import types
from contextlib import redirect_stdout
from pathlib import Path

import numpy as np

SRC = Path(__file__).resolve().parent          # this directory: scenarios.py lives here
HERE = SRC                                      # where the vectors are written (main() may redirect it: --out DIR)
ROOT = SRC.parent.parent
REF = Path("/root/reference/02_Visual_Engine")
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(REF))

from omfs_4d_video_gen_amd.engine import synthetic  # noqa: E402


def tiny_png(path, value, size=8):
    from PIL import Image
    Image.fromarray(np.full((size, size, 3), value, dtype=np.uint8)).save(path)


def make_fixture_dataset(root: Path, n_frames=60, with_masks=False, gaps=0):
    """Small dataset in the reference layout (preprocess_video.py:314-416); byte-stable."""
    root.mkdir(parents=True, exist_ok=True)
    (root / "images").mkdir(exist_ok=True)
    (root / "flame_param").mkdir(exist_ok=True)
    seq = synthetic.make_flame_sequence(n_frames, seed=11)
    np.savez(root / "flame_param.npz", **seq)
    frames = []
    t_idx = 0
    for i in range(n_frames):
        tiny_png(root / "images" / f"{i:05d}_00.png", (i * 3) % 255)
        per = {k: (v[i:i + 1] if v.ndim > 1 and v.shape[0] == n_frames else v) for k, v in seq.items()}
        np.savez(root / "flame_param" / f"{i:05d}.npz", **per)
        frames.append({"file_path": f"images/{i:05d}_00.png", "flame_param_path": f"flame_param/{i:05d}.npz",
                       "transform_matrix": [[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 1], [0, 0, 0, 1]],
                       "timestep_index": t_idx, "camera_index": 0, "camera_angle_x": 0.5, "w": 8, "h": 8})
        t_idx += 3 if (gaps and i % max(1, n_frames // gaps) == 0) else 1
    if with_masks:
        (root / "fg_masks").mkdir(exist_ok=True)
        for i in range(n_frames):
            tiny_png(root / "fg_masks" / f"{i:05d}_00.png", 255)
    top = {"camera_angle_x": 0.5, "camera_angle_y": 0.5, "fl_x": 15.6, "fl_y": 15.6, "cx": 4.0, "cy": 4.0, "w": 8, "h": 8,
           "timestep_indices": list(range(n_frames)), "camera_indices": [0]}
    split = max(1, n_frames - n_frames // 10)
    for name, fr in (("transforms_train.json", frames[:split]), ("transforms_test.json", frames[split:]),
                     ("transforms_val.json", frames[split:])):
        with open(root / name, "w") as f:
            json.dump({**top, "frames": fr}, f, indent=2)
    canon = {k: (np.zeros((1,) + v.shape[1:], v.dtype) if v.ndim > 1 and v.shape[0] == n_frames else v) for k, v in seq.items()}
    np.savez(root / "canonical_flame_param.npz", **canon)


class _Monkey:
    """setattr with restore (what pytest's monkeypatch does for the tests that replay the scenarios)."""

    def __init__(self):
        self.undo = []

    def __call__(self, obj, name, value):
        self.undo.append((obj, name, getattr(obj, name)))
        setattr(obj, name, value)

    def restore(self):
        for obj, name, old in reversed(self.undo):
            setattr(obj, name, old)
        self.undo.clear()


def golden_render_surgery(out):
    rs = importlib.import_module("render_surgery")
    grid = [(mm, s) for mm in (-7.5, -1.0, 0.0, 0.25, 3.0, 12.0) for s in (0.0, 0.5, 1.0, 2.5)]
    out["compute_offset"] = {"inputs": grid, "outputs": [rs.compute_offset(mm, s) for mm, s in grid],
                             "SCALE_FACTOR": rs.SCALE_FACTOR}
    tmp = Path(tempfile.mkdtemp())
    rng = np.random.default_rng(5)
    cases = {}
    base2d = {"jaw_pose": rng.standard_normal((6, 3)).astype(np.float32), "translation": rng.standard_normal((6, 3)).astype(np.float32),
              "expr": rng.standard_normal((6, 100)).astype(np.float32), "shape": rng.standard_normal(300).astype(np.float32)}
    base1d = {"jaw_pose": base2d["jaw_pose"][0], "translation": base2d["translation"][0], "expr": base2d["expr"][0], "shape": base2d["shape"]}
    for name, base in (("2d", base2d), ("1d", base1d)):
        for dname, dmap in (("default", None), ("map", {"translation_axis": 2, "jaw_axis": 1, "lefort_scale": 2.0, "bsso_scale": 0.5})):
            src, dst = tmp / f"{name}_{dname}_src.npz", tmp / f"{name}_{dname}_dst.npz"
            np.savez(src, **base)
            rs.modify_flame_params(str(src), str(dst), 0.0045, -0.0021, deformation_map=dmap)
            res = dict(np.load(dst))
            cases[f"{name}_{dname}"] = {"deformation_map": dmap}
            for k, v in res.items():
                np.save(HERE / f"rs_modify_{name}_{dname}_{k}.npy", v)
    np.savez(HERE / "rs_modify_inputs.npz", **{f"2d_{k}": v for k, v in base2d.items()}, **{f"1d_{k}": v for k, v in base1d.items()})
    out["modify_flame_params"] = {"lefort_offset": 0.0045, "bsso_offset": -0.0021, "cases": cases}
    asset = tmp / "asset.npz"
    np.savez(asset, version=np.array([1]))
    out["choose_rig_mode"] = [
        {"args": ["flame_only", ""], "result": list(rs.choose_rig_mode("flame_only", ""))},
        {"args": ["flame_only", "<existing>"], "result": list(rs.choose_rig_mode("flame_only", str(asset)))},
        {"args": ["hybrid_full_head", ""], "result": list(rs.choose_rig_mode("hybrid_full_head", ""))},
        {"args": ["hybrid_full_head", "<missing>"], "result": list(rs.choose_rig_mode("hybrid_full_head", str(tmp / "nope.npz")))},
        {"args": ["hybrid_full_head", "<existing>"], "result": list(rs.choose_rig_mode("hybrid_full_head", str(asset)))},
    ]
    sel = {}
    for n in (1, 2, 6, 24, 25, 300):
        fd, od = tmp / f"frames_{n}", tmp / f"out_{n}"
        fd.mkdir()
        for i in range(n):
            tiny_png(fd / f"{i:05d}.png", i % 255, size=2)
        with redirect_stdout(io.StringIO()):
            rs.export_deterministic_frames(str(fd), str(od), None, 24)
        man = json.loads((od / "deterministic_indices_manifest.json").read_text())
        sel[str(n)] = {"selected_indices": man["selected_indices"], "exports": man["exports"]}
    fd = tmp / "frames_300"
    with redirect_stdout(io.StringIO()):
        rs.export_deterministic_frames(str(fd), str(tmp / "out_300_7"), None, 7)
    sel["300_max7"] = {"selected_indices": json.loads((tmp / "out_300_7" / "deterministic_indices_manifest.json").read_text())["selected_indices"]}
    out["export_deterministic_frames"] = sel
    ds = tmp / "dataset"
    make_fixture_dataset(ds, 12)
    with redirect_stdout(io.StringIO()):
        mod = Path(rs.create_modified_dataset(str(ds), 0.003, 0.001))
    listing = sorted(str(p.relative_to(mod)) for p in mod.rglob("*") if p.is_file() or p.is_symlink())
    tj = json.loads((mod / "transforms_train.json").read_text())
    before = np.load(ds / "flame_param" / "00003.npz")
    after = np.load(mod / "flame_param" / "00003.npz")
    out["create_modified_dataset"] = {
        "lefort_offset": 0.003, "bsso_offset": 0.001, "files": listing,
        "frame0": tj["frames"][0], "n_train_frames": len(tj["frames"]),
        "per_frame_translation_delta": (after["translation"] - before["translation"]).tolist(),
        "per_frame_jaw_delta": (after["jaw_pose"] - before["jaw_pose"]).tolist(),
        "batched_translation_delta_row0": (np.load(mod / "flame_param.npz")["translation"][0] - np.load(ds / "flame_param.npz")["translation"][0]).tolist()}
    shutil.rmtree(mod, ignore_errors=True)
    shutil.rmtree(tmp, ignore_errors=True)


def golden_train_ghost(out):
    tg = importlib.import_module("train_ghost")
    tmp = Path(tempfile.mkdtemp())
    gates = {}
    for name, kw in (("ok_60", dict(n_frames=60)), ("too_few_40", dict(n_frames=40)), ("gappy_100", dict(n_frames=100, gaps=30)),
                     ("masks_ok_60", dict(n_frames=60, with_masks=True))):
        d = tmp / name
        make_fixture_dataset(d, **kw)
        try:
            with redirect_stdout(io.StringIO()) as buf:
                tg.run_quality_gates(str(d))
            gates[name] = {"ok": True, "stdout": buf.getvalue().strip()}
        except RuntimeError as e:
            gates[name] = {"ok": False, "error": str(e)}
    d = tmp / "few_masks_60"
    make_fixture_dataset(d, n_frames=60, with_masks=True)
    for p in sorted((d / "fg_masks").iterdir())[10:]:
        p.unlink()
    try:
        with redirect_stdout(io.StringIO()):
            tg.run_quality_gates(str(d))
        gates["few_masks_60"] = {"ok": True}
    except RuntimeError as e:
        gates["few_masks_60"] = {"ok": False, "error": str(e)}
    out["run_quality_gates"] = gates
    # fingerprint of a committed, byte-stable mini dataset
    fp_dir = HERE / "fingerprint_dataset"
    if fp_dir.exists():
        shutil.rmtree(fp_dir)
    fp_dir.mkdir()
    for name, payload in (("transforms_train.json", {"frames": [{"file_path": "images/00000_00.png"}]}),
                          ("transforms_test.json", {"frames": []}), ("transforms_val.json", {"frames": []})):
        (fp_dir / name).write_text(json.dumps(payload, indent=2))
    (fp_dir / "flame_param.npz").write_bytes(b"golden-bytes-flame-param")
    out["build_dataset_fingerprint"] = tg.build_dataset_fingerprint(str(fp_dir))
    # argv / save-iteration rule / manifest schema of train(): capture the engine launch
    captured = {}

    class FakeResult:
        returncode = 0

    def fake_run(cmd, **kw):
        captured.setdefault("calls", []).append({"cmd": list(cmd), "cwd": kw.get("cwd"), "has_pythonpath": "PYTHONPATH" in kw.get("env", {}),
                                                 "capture_output": kw.get("capture_output"), "text": kw.get("text")})
        return FakeResult()

    # every patch of the reference's module goes through _Monkey and is undone before the next section runs: a leaked
    # `validate_setup = lambda: None` made the later `surface` section record "returned None" for the two refusals
    monkey = _Monkey()
    monkey(tg, "validate_setup", lambda: None)
    monkey(tg.subprocess, "run", fake_run)
    argv = {}
    for iters, masks in ((3000, False), (5000, False), (30000, True), (600000, False)):
        d = tmp / f"train_{iters}"
        make_fixture_dataset(d, n_frames=60, with_masks=masks)
        outdir = tmp / f"model_{iters}"
        try:
            with redirect_stdout(io.StringIO()) as buf:
                tg.train(str(d), str(outdir), iterations=iters, resolution=-1)
        except BaseException:
            monkey.restore()
            raise
        cmd = captured["calls"][-1]["cmd"]
        cmd_rel = [c.replace(str(d.resolve()), "<DATA>").replace(str(outdir.resolve()), "<MODEL>").replace(str(tg.REPO_DIR), "<ENGINE>") for c in cmd[1:]]
        man = json.loads(next((outdir / "experiment_manifests").iterdir()).read_text())
        argv[str(iters)] = {"argv_after_python": cmd_rel, "has_masks": masks, "manifest_keys": sorted(man.keys()),
                            "manifest_extra": man["extra"], "fingerprint_keys": sorted(man["dataset_fingerprint"].keys()),
                            "kw": {k: captured["calls"][-1][k] for k in ("has_pythonpath", "capture_output", "text")},
                            "stdout_lines": [l.replace(str(tmp), "<TMP>") for l in buf.getvalue().splitlines()
                                             if l.startswith("[train_ghost]") and "manifest" not in l and "Command" not in l]}
    out["train_argv"] = argv

    class FailResult:
        returncode = 3
    monkey(tg.subprocess, "run", lambda cmd, **kw: FailResult())
    d = tmp / "train_fail"
    make_fixture_dataset(d, n_frames=60)
    try:
        with redirect_stdout(io.StringIO()):
            tg.train(str(d), str(tmp / "model_fail"), iterations=100)
    except RuntimeError as e:
        out["train_failure_message"] = str(e)
    finally:
        monkey.restore()
        shutil.rmtree(tmp, ignore_errors=True)


def golden_flame_fitter(out):
    for name in ("cv2", "mediapipe"):
        sys.modules.setdefault(name, types.ModuleType(name))
    ff = importlib.import_module("flame_fitter")
    import torch
    tmp = Path(tempfile.mkdtemp())
    rig = synthetic.make_rig(seed=0)
    pkl, lmk = tmp / "flame2023.pkl", tmp / "landmark_embedding_with_eyes.npy"
    synthetic.write_flame_pickle(rig, str(pkl), str(lmk))
    ff.FLAME_LMK_PATH = lmk
    torch.manual_seed(0)
    model = ff.SimpleFLAME(str(pkl), n_shape=100, n_expr=50)
    rng = np.random.default_rng(21)
    aa = np.concatenate([rng.standard_normal((6, 3)).astype(np.float32) * 0.7, np.zeros((1, 3), np.float32),
                         np.array([[1e-9, 0, 0]], np.float32)])
    R = model._axis_angle_to_matrix(torch.from_numpy(aa)).numpy()
    B = 5
    shape = (rng.standard_normal((1, 100)) * 0.5).astype(np.float32).repeat(B, 0)
    expr = (rng.standard_normal((B, 50)) * 0.5).astype(np.float32)
    rot = (rng.standard_normal((B, 3)) * 0.3).astype(np.float32)
    jaw = np.abs(rng.standard_normal((B, 3)) * 0.2).astype(np.float32)
    trans = (rng.standard_normal((B, 3)) * 0.05 + np.array([0, 0, -5.0])).astype(np.float32)
    with torch.no_grad():
        lm = model(*[torch.from_numpy(a) for a in (shape, expr, rot, jaw, trans)]).numpy()
    # synthetic 2-D landmarks: project a ground-truth sequence with the fitter's own camera model
    T, W, H = 6, 512, 512
    gt_expr = (rng.standard_normal((T, 50)) * 0.4).astype(np.float32)
    gt_rot = (rng.standard_normal((T, 3)) * 0.1).astype(np.float32)
    gt_jaw = np.abs(rng.standard_normal((T, 3)) * 0.1).astype(np.float32)
    gt_trans = np.tile(np.array([[0.02, -0.01, -5.0]], np.float32), (T, 1))
    with torch.no_grad():
        l3 = model(torch.zeros(T, 100), torch.from_numpy(gt_expr), torch.from_numpy(gt_rot), torch.from_numpy(gt_jaw), torch.from_numpy(gt_trans)).numpy()
    px = (l3[:, :, 0] / (-l3[:, :, 2] + 1e-8) + 1) * 0.5 * W
    py = (l3[:, :, 1] / (-l3[:, :, 2] + 1e-8) + 1) * 0.5 * H
    lmk2d = [np.stack([px[t], py[t]], -1).astype(np.float32) for t in range(T)]
    lmk2d[3] = None   # a frame without a detected face
    poses = [list(ff.estimate_head_pose_from_landmarks(l, (W, H))) for l in lmk2d]
    fits, fit_stdout = {}, {}
    for iters in (1, 3, 200):
        with redirect_stdout(io.StringIO()) as sink:
            # 200 is the reference's default fit length (flame_fitter.py:302): that run is made WITHOUT the argument
            kw = {} if iters == 200 else {"n_iters": iters}
            res = ff.fit_flame_to_landmarks([None if l is None else l.copy() for l in lmk2d], (W, H), str(pkl), n_shape=100, n_expr=50,
                                            lr=0.01, device="cpu", **kw)
        fits[iters] = res
        fit_stdout[str(iters)] = sink.getvalue().splitlines()
    np.savez_compressed(
        HERE / "flame_fitter_golden.npz", rig_seed=np.array([0]), axis_angle=aa, rotmats=R,
        fwd_shape=shape, fwd_expr=expr, fwd_rot=rot, fwd_jaw=jaw, fwd_trans=trans, fwd_landmarks=lm,
        lmk2d=np.stack([np.zeros((68, 2), np.float32) if l is None else l for l in lmk2d]),
        lmk2d_valid=np.array([l is not None for l in lmk2d]), image_size=np.array([W, H]), head_pose_init=np.array(poses, np.float32),
        **{f"fit{it}_{k}": v for it, r in fits.items() for k, v in r.items() if k not in ("static_offset", "dynamic_offset")},
        fit_static_offset_shape=np.array(fits[3]["static_offset"].shape), fit_dynamic_offset_shape=np.array(fits[3]["dynamic_offset"].shape))
    sys.path.insert(0, str(SRC))
    import scenarios as SC
    (tmp / "detect").mkdir()
    detect = SC.detect_landmarks(ff, tmp / "detect")
    (tmp / "fv").mkdir()
    monkey = _Monkey()
    try:
        fv = SC.fit_video(ff, tmp / "fv", monkey, str(pkl), str(lmk), "cpu", lmk2d, (W, H))
    finally:
        monkey.restore()
    fv_arrays = fv.pop("_arrays")
    np.savez_compressed(HERE / "flame_fitter_fit_video_golden.npz", **fv_arrays)
    out["flame_fitter"] = {"detect_landmarks": detect, "fit_video": fv, "fit_stdout": fit_stdout, "result_keys": sorted(fits[3].keys()), "n_landmarks": int(lm.shape[1]),
                           "shapes": {k: list(v.shape) for k, v in fits[3].items()}}
    shutil.rmtree(tmp, ignore_errors=True)


def make_vhap_export(root: Path, n_frames=11):
    """Synthetic VHAP export (what `preprocess_video.py:212-300` reads); byte-stable."""
    (root / "images").mkdir(parents=True, exist_ok=True)
    (root / "fg_masks").mkdir(exist_ok=True)
    (root / "flame_param").mkdir(exist_ok=True)
    seq = synthetic.make_flame_sequence(n_frames, seed=31)
    frames = []
    for i in range(n_frames):
        tiny_png(root / "images" / f"{i:05d}_00.png", i * 7 % 255)
        tiny_png(root / "fg_masks" / f"{i:05d}_00.png", 255)
        per = {k: (v[i:i + 1] if v.ndim > 1 and v.shape[0] == n_frames else v) for k, v in seq.items()}
        if i == 4:
            continue                      # a frame whose FLAME file is missing is skipped in the batched file
        np.savez(root / "flame_param" / f"{i:05d}.npz", **per)
    for i in range(n_frames):
        fr = {"file_path": f"images/{i:05d}_00.png", "fg_mask_path": f"fg_masks/{i:05d}_00.png",
              "flame_param_path": f"flame_param/{i:05d}.npz", "timestep_index": i, "camera_index": 0,
              "transform_matrix": [[1, 0, 0, 0.01 * i], [0, 1, 0, 0], [0, 0, 1, 1], [0, 0, 0, 1]]}
        if i == 0:
            fr.update({"fl_x": 1234.5, "fl_y": 1230.0, "cx": 256.0, "cy": 200.0, "w": 512, "h": 400})
        if i == 2:
            del fr["fg_mask_path"]
        frames.append(fr)
    (root / "transforms.json").write_text(json.dumps({"camera_angle_x": 9.9, "frames": frames}, indent=2))


def golden_preprocess(out):
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    pv = importlib.import_module("preprocess_video")
    tmp = Path(tempfile.mkdtemp())
    make_vhap_export(tmp / "export")
    with redirect_stdout(io.StringIO()):
        res = pv.convert_to_gaussianavatars_format(tmp / "export", tmp / "out")
    o = tmp / "out"
    out["convert"] = {"result": {**res, "output_dir": "<OUT>", "image_size": list(res["image_size"])},
                      "files": sorted(str(p.relative_to(o)) for p in o.rglob("*") if p.is_file()),
                      "transforms_train": json.loads((o / "transforms_train.json").read_text()),
                      "n_test": len(json.loads((o / "transforms_test.json").read_text())["frames"]),
                      "val_equals_test": (o / "transforms_val.json").read_text() == (o / "transforms_test.json").read_text()}
    b, c = np.load(o / "flame_param.npz"), np.load(o / "canonical_flame_param.npz")
    np.savez_compressed(HERE / "preprocess_convert_golden.npz", **{f"batched_{k}": b[k] for k in b.files if k != "dynamic_offset"},
                        batched_dynamic_offset_shape=np.array(b["dynamic_offset"].shape), batched_dynamic_offset_absmax=np.abs(b["dynamic_offset"]).max(),
                        **{f"canonical_{k}_shape": np.array(c[k].shape) for k in c.files},
                        canonical_nonzero=np.array([float(np.abs(c[k]).max()) for k in sorted(c.files) if k not in ("shape", "static_offset")]))
    shutil.rmtree(tmp, ignore_errors=True)


def golden_surface(out):
    """The rest of the importable call surface (Missing 1), through the shared scenario drivers."""
    sys.path.insert(0, str(SRC))
    import scenarios as SC
    vr = importlib.import_module("validation_reporting")
    rs = importlib.import_module("render_surgery")
    sfe = importlib.import_module("single_frame_experiment")
    hr = {n: importlib.import_module(f"head_recon.{n}") for n in ("ingest_sequences", "register_sequences", "build_canonical_head", "eval_head_coverage")}
    for m in (vr, rs, sfe, *hr.values()):
        assert str(REF) in os.path.abspath(m.__file__), m.__file__       # the REFERENCE's modules, not the drop-ins
    tmp = Path(tempfile.mkdtemp())
    monkey = _Monkey()
    try:
        surf = {"validation_metrics": SC.validation_metrics(vr), "validation_report": SC.validation_report(vr, tmp / "vr")}
        (tmp / "hr").mkdir()
        surf["head_recon"] = SC.head_recon(hr["ingest_sequences"], hr["register_sequences"], hr["build_canonical_head"], hr["eval_head_coverage"], tmp / "hr")
        (tmp / "launch").mkdir()
        surf["render_launch"] = SC.render_launch(rs, tmp / "launch", monkey)
        monkey.restore()
        (tmp / "stitch").mkdir()
        surf["stitch_video"] = SC.stitch(rs, tmp / "stitch", monkey)
        monkey.restore()
        (tmp / "dmap").mkdir()
        surf["load_deformation_map"] = SC.deformation_map(rs, tmp / "dmap")

        def build(src, dst):
            monkey(sfe, "DATA_CONDA", src)
            monkey(sfe, "DATA_SINGLE", dst)
            return sfe.build_single_frame_dataset()
        (tmp / "sf").mkdir()
        surf["single_frame_dataset"] = SC.single_frame_dataset(build, make_fixture_dataset, tmp / "sf")
        monkey.restore()
        tg = importlib.import_module("train_ghost")
        assert str(REF) in os.path.abspath(tg.__file__)
        (tmp / "rs_cli").mkdir()
        surf["render_surgery_cli"] = SC.render_surgery_cli(rs, make_fixture_dataset, tmp / "rs_cli", monkey)
        monkey.restore()
        (tmp / "tg_cli").mkdir()
        surf["train_ghost_cli"] = SC.train_ghost_cli(tg, make_fixture_dataset, tmp / "tg_cli", monkey)
        monkey.restore()
        (tmp / "vr_cli").mkdir()
        surf["validation_reporting_cli"] = SC.validation_reporting_cli(vr, tmp / "vr_cli")
        (tmp / "helpers").mkdir()
        surf["helper_functions"] = SC.helper_functions(rs, tg, make_fixture_dataset, tmp / "helpers", monkey)
        monkey.restore()
    finally:
        monkey.restore()
        shutil.rmtree(tmp, ignore_errors=True)
    out["surface"] = surf


SECTIONS = {"preprocess": golden_preprocess, "render_surgery": golden_render_surgery, "train_ghost": golden_train_ghost,
            "flame_fitter": golden_flame_fitter, "surface": golden_surface}


def main():
    global HERE
    args = sys.argv[1:]
    if "--out" in args:                          # tests/test_reference_goldens.py::test_generator_reproduces_the_committed_goldens
        i = args.index("--out")
        HERE = Path(args[i + 1]).resolve()
        HERE.mkdir(parents=True, exist_ok=True)
        del args[i:i + 2]
    target = HERE / "reference_goldens.json"
    wanted = args or list(SECTIONS)
    out = json.loads(target.read_text()) if target.exists() and args else {}
    for name in wanted:
        SECTIONS[name](out)
    with open(target, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("goldens written to", HERE, "sections:", wanted)


if __name__ == "__main__":
    main()

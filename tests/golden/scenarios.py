"""Scenario drivers shared by `make_goldens.py` (which hands them the REFERENCE's modules, imported from
/root/reference in the build container) and `tests/test_reference_goldens.py` (which hands them this
repo's drop-in modules): the same seeded inputs, the same stubs for the child processes, the same
normalisation of temporary paths -- so a golden is "what the reference did in this scenario" and the test
is "the drop-in does the same".  Test infrastructure only; nothing here reads /root/reference.

Every driver returns JSON-serialisable data (floats as Python floats: json round-trips them exactly).
"""
from __future__ import annotations

import io
import json
import os
import struct
import subprocess
import sys
import zlib
from contextlib import redirect_stdout
from pathlib import Path

import numpy as np


# ------------------------------------------------------------------ small helpers
def _png_bytes(img: np.ndarray) -> bytes:
    """8-bit gray / RGB / RGBA PNG without any imaging dependency (filter 0, zlib level 6): byte-stable."""
    a = np.ascontiguousarray(img, np.uint8)
    if a.ndim == 2:
        a = a[:, :, None]
    h, w, c = a.shape
    raw = np.zeros((h, 1 + w * c), np.uint8)
    raw[:, 1:] = a.reshape(h, w * c)

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, {1: 0, 3: 2, 4: 6}[c], 0, 0, 0))
            + chunk(b"IDAT", zlib.compress(raw.tobytes(), 6)) + chunk(b"IEND", b""))


def put_png(path: Path, img: np.ndarray) -> None:
    Path(path).write_bytes(_png_bytes(img))


def _norm(text: str, subs: dict) -> str:
    """Replace temporary paths by stable tokens, longest path first."""
    for real in sorted(subs, key=len, reverse=True):
        text = text.replace(real, subs[real])
    return text


def _call(fn, *a, subs=None, **kw):
    """Run fn, capture stdout and the exception (type + message), normalise paths."""
    subs = subs or {}
    buf = io.StringIO()
    rec = {}
    try:
        with redirect_stdout(buf):
            rec["returned"] = fn(*a, **kw)
    except Exception as e:  # noqa: BLE001 -- the exception IS the datum
        rec["raised"] = [type(e).__name__, _norm(str(e), subs)]
    rec["stdout"] = [_norm(l, subs) for l in buf.getvalue().splitlines()]
    return rec


def _tree(root: Path) -> list:
    root = Path(root)
    if not root.exists():
        return []
    return sorted(str(p.relative_to(root)) + ("/" if p.is_dir() else "") for p in root.rglob("*"))


# ------------------------------------------------------------------ validation_reporting (reference validation_reporting.py:16-123)
def validation_metrics(vr) -> dict:
    rng = np.random.default_rng(77)
    a = rng.integers(0, 256, (14, 11, 3)).astype(np.float32)
    noisy = np.clip(a + rng.normal(0, 9, a.shape), 0, 255).astype(np.float32)
    grey = np.full((14, 11, 3), 128.0, np.float32)
    grey2 = np.full((14, 11, 3), 131.0, np.float32)
    a2d, n2d = a[:, :, 0].copy(), noisy[:, :, 1].copy()
    u8a, u8b = a.astype(np.uint8), noisy.astype(np.uint8)             # uint8 inputs wrap in (a - b): captured as the reference does it
    pairs = {"identical": (a, a), "noisy": (a, noisy), "grey_equal": (grey, grey), "grey_offset": (grey, grey2),
             "two_dim": (a2d, n2d), "inverted": (a, 255.0 - a), "float64": (a.astype(np.float64), noisy.astype(np.float64)),
             "uint8": (u8a, u8b), "black_white": (np.zeros((6, 5, 3), np.float32), np.full((6, 5, 3), 255.0, np.float32))}
    out = {"psnr": {}, "ssim_global": {}}
    for name, (x, y) in pairs.items():
        out["psnr"][name] = vr.psnr(x, y)
        out["ssim_global"][name] = vr.ssim_global(x, y)
    grid = [-0.1, 0.0, 0.1, 0.19999, 0.2, 0.20001, 0.3, 0.34999, 0.35, 0.5, 0.65, 0.65001, 0.7, 0.79999, 0.8, 0.80001, 1.0, 1.5]
    out["bucket"] = [[p, vr._bucket(p)] for p in grid]
    return out


def make_report_tree(root: Path) -> dict:
    """<root>/model/train/ours_{7,30,100}/{renders,gt} + <root>/det/deterministic_indices_manifest.json; byte-stable."""
    rng = np.random.default_rng(123)
    model, det = root / "model", root / "det"
    for it in (7, 30, 100):
        for sub in ("renders", "gt"):
            (model / "train" / f"ours_{it}" / sub).mkdir(parents=True, exist_ok=True)
    (model / "train" / "not_ours_999").mkdir()
    (model / "train" / "ours_note.txt").write_text("a file, not a run directory")
    names = [f"{i:05d}.png" for i in range(12)]
    for i, name in enumerate(names):
        gt = rng.integers(0, 256, (10, 12, 3)).astype(np.uint8)
        rd = np.clip(gt.astype(np.int32) + rng.integers(-20, 21, gt.shape), 0, 255).astype(np.uint8)
        if i == 2:
            rd = gt.copy()                                             # identical pair: PSNR 99
        if i == 4:                                                     # RGBA render, gray ground truth
            rd = np.concatenate([rd, np.full((10, 12, 1), 77, np.uint8)], 2)
            gt = gt[:, :, 0]
        for it, shift in ((100, 0), (30, 40)):                         # the older run holds different pixels
            put_png(model / "train" / f"ours_{it}" / "renders" / name, np.clip(rd.astype(np.int32) + shift, 0, 255).astype(np.uint8))
            put_png(model / "train" / f"ours_{it}" / "gt" / name, gt)
    (model / "train" / "ours_100" / "renders" / names[9]).unlink()     # a render that is missing: row skipped
    (model / "train" / "ours_100" / "gt" / names[10]).unlink()         # a ground truth that is missing: row skipped
    det.mkdir(parents=True, exist_ok=True)
    rows = [{"index": i, "source": names[i], "exported": f"idx_{i:05d}.png"} for i in (0, 2, 3, 4, 5, 7, 8, 9, 10, 11)]
    (det / "deterministic_indices_manifest.json").write_text(json.dumps(
        {"source_frames_dir": "<renders>", "selected_indices": [r["index"] for r in rows], "exports": rows}, indent=2))
    return {"model": model, "det": det}


def validation_report(vr, tmp: Path) -> dict:
    tmp = Path(tmp)
    t = make_report_tree(tmp / "tree")
    subs = {str(tmp): "<TMP>"}
    out = {}
    rec = _call(vr.generate_report, t["model"], t["det"], tmp / "rep", subs=subs)
    rec.pop("returned", None)
    out["ok"] = {**rec, "strict_scores": json.loads((tmp / "rep" / "strict_scores.json").read_text()),
                 "checklist": (tmp / "rep" / "human_review_checklist.md").read_text(), "files": _tree(tmp / "rep")}
    # an empty export list, and a manifest without the key
    (tmp / "det_empty").mkdir()
    (tmp / "det_empty" / "deterministic_indices_manifest.json").write_text(json.dumps({"exports": []}))
    _call(vr.generate_report, t["model"], tmp / "det_empty", tmp / "rep_empty", subs=subs)
    out["empty_exports"] = json.loads((tmp / "rep_empty" / "strict_scores.json").read_text())
    (tmp / "det_one").mkdir()
    (tmp / "det_one" / "deterministic_indices_manifest.json").write_text(json.dumps(
        {"exports": [{"index": 0, "source": "00000.png", "exported": "idx_00000.png"}]}))
    _call(vr.generate_report, t["model"], tmp / "det_one", tmp / "rep_one", subs=subs)
    out["single_row_index0"] = json.loads((tmp / "rep_one" / "strict_scores.json").read_text())
    # the four refusals
    errs = {}
    errs["no_train_dir"] = _call(vr.generate_report, tmp / "nowhere", t["det"], tmp / "r1", subs=subs)
    (tmp / "bare" / "train" / "other").mkdir(parents=True)
    errs["no_ours_dirs"] = _call(vr.generate_report, tmp / "bare", t["det"], tmp / "r2", subs=subs)
    (tmp / "nogt" / "train" / "ours_5" / "renders").mkdir(parents=True)
    errs["no_gt"] = _call(vr.generate_report, tmp / "nogt", t["det"], tmp / "r3", subs=subs)
    (tmp / "det_missing").mkdir()
    errs["no_manifest"] = _call(vr.generate_report, t["model"], tmp / "det_missing", tmp / "r4", subs=subs)
    for v in errs.values():
        v.pop("returned", None)
    out["errors"] = errs
    return out


# ------------------------------------------------------------------ head_recon (reference head_recon/*.py)
def make_capture_root(root: Path) -> Path:
    cap = root / "captures"
    layout = {"seq_b_front": (["0.png", "1.PNG", "2.jpg", "3.JPG", "notes.txt", "4.jpeg"], True),
              "seq_a_left": (["a.png", "b.png"], False), "seq_c_transforms_only": (None, True), "seq_d_empty": (None, False)}
    for name, (images, transforms) in layout.items():
        (cap / name).mkdir(parents=True)
        if images is not None:
            (cap / name / "images").mkdir()
            for f in images:
                (cap / name / "images" / f).write_bytes(b"x")
        if transforms:
            (cap / name / "transforms_train.json").write_text(json.dumps({"frames": [{}] * 3}))
    (cap / "stray_file.json").write_text("{}")
    return cap


def head_recon(ingest, register, build, coverage, tmp: Path) -> dict:
    """ingest / register / build / coverage: the four modules (reference's or the drop-in's)."""
    tmp = Path(tmp)
    subs = {str(tmp.resolve()): "<TMP>", str(tmp): "<TMP>"}
    cap = make_capture_root(tmp)
    out_dir = tmp / "out" / "head_recon"
    out = {}
    r = _call(ingest.ingest_sequences, cap, out_dir, subs=subs)
    r["returned"] = _norm(str(r["returned"]), subs)
    out["ingest"] = {**r, "manifest_text": _norm((out_dir / "sequence_manifest.json").read_text(), subs)}
    r = _call(register.register_sequences, out_dir / "sequence_manifest.json", out_dir, subs=subs)
    r["returned"] = _norm(str(r["returned"]), subs)
    out["register"] = {**r, "registration_text": _norm((out_dir / "registration.json").read_text(), subs)}
    r = _call(build.build_canonical_head, out_dir / "registration.json", out_dir, subs=subs)
    r["returned"] = [_norm(str(p), subs) for p in r["returned"]]
    asset = np.load(out_dir / "canonical_head_asset.npz")
    out["build"] = {**r, "manifest_text": _norm((out_dir / "canonical_head_asset_manifest.json").read_text(), subs),
                    "asset": {k: {"dtype": str(asset[k].dtype), "value": asset[k].tolist()} for k in sorted(asset.files)}}
    out["files"] = _tree(out_dir)
    # a registration file without the optional keys
    (tmp / "reg_bare.json").write_text("{}")
    r = _call(build.build_canonical_head, tmp / "reg_bare.json", tmp / "out_bare", subs=subs)
    a2 = np.load(tmp / "out_bare" / "canonical_head_asset.npz")
    out["build_bare"] = {"asset": {k: a2[k].tolist() for k in sorted(a2.files)},
                         "manifest_text": _norm((tmp / "out_bare" / "canonical_head_asset_manifest.json").read_text(), subs)}
    # refusals
    (tmp / "empty_manifest.json").write_text(json.dumps({"sequences": []}))
    (tmp / "no_key_manifest.json").write_text("{}")
    out["register_empty"] = _call(register.register_sequences, tmp / "empty_manifest.json", tmp / "o2", subs=subs)
    out["register_no_key"] = _call(register.register_sequences, tmp / "no_key_manifest.json", tmp / "o3", subs=subs)
    out["register_wrote_nothing"] = not (tmp / "o2").exists() and not (tmp / "o3").exists()
    out["coverage"] = {str(n): coverage.evaluate_head_coverage(n) for n in (-3, 0, 1, 2, 3, 5, 6, 10, 21, 100, 101, 300)}
    # the coverage CLI
    (tmp / "cov_transforms.json").write_text(json.dumps({"frames": [{"i": i} for i in range(17)]}))
    argv = sys.argv
    sys.argv = ["eval_head_coverage", "--transforms", str(tmp / "cov_transforms.json"), "--output", str(tmp / "cov" / "deep" / "head_coverage.json")]
    try:
        r = _call(coverage.main, subs=subs)
    finally:
        sys.argv = argv
    r.pop("returned", None)
    out["coverage_cli"] = {**r, "text": (tmp / "cov" / "deep" / "head_coverage.json").read_text()}
    return out


# ------------------------------------------------------------------ render_surgery: engine launch, discovery, ffmpeg (reference :58-71, :245-362, :412-449)
class _Result:
    def __init__(self, returncode=0, stdout="", stderr=""):
        self.returncode, self.stdout, self.stderr = returncode, stdout, stderr


def _model_dir(root: Path, point_clouds, stale) -> Path:
    model = root / "model"
    model.mkdir(parents=True)
    for name in point_clouds:
        (model / "point_cloud" / name).mkdir(parents=True)
    for it, sub in stale:
        (model / "train" / f"ours_{it}" / sub).mkdir(parents=True)
        put_png(model / "train" / f"ours_{it}" / sub / "00000.png", np.zeros((2, 2, 3), np.uint8))
    return model


def render_launch(rs, tmp: Path, monkey) -> dict:
    """`monkey(obj, name, value)` sets an attribute for the duration of the scenario (pytest's monkeypatch.setattr or a
    plain setattr with restore)."""
    tmp = Path(tmp)
    engine = tmp / "engine"
    engine.mkdir()
    (engine / "render.py").write_text("# placeholder engine entry point\n")
    monkey(rs, "REPO_DIR", engine)
    monkey(rs, "RENDER_SCRIPT", engine / "render.py")
    data = tmp / "data"
    data.mkdir()
    out = {}
    cases = {
        # name: (iteration, clear_old, point-cloud dirs, stale (iteration, subdir) pairs, what the child writes, child rc)
        "auto_no_point_clouds": (-1, True, [], [(5, "renders"), (30, "renders"), (99, "gt")], {12: 3}, 0),
        "auto_with_point_clouds": (-1, True, ["iteration_5", "iteration_30", "iteration_x", "iteration_", "other_40", "iteration_7_b"],
                                   [(5, "renders"), (40, "renders")], {30: 4}, 0),
        "zero_means_auto": (0, True, ["iteration_5"], [], {5: 1}, 0),
        "pinned_7_no_point_clouds": (7, True, [], [(30, "renders")], {7: 2}, 0),
        "pinned_7_with_point_clouds": (7, True, ["iteration_5", "iteration_30"], [(30, "renders"), (7, "renders")], {7: 2}, 0),
        "pinned_7_child_wrote_9_and_30": (7, True, ["iteration_30"], [], {9: 2, 30: 5}, 0),
        "keep_old_renders": (-1, False, ["iteration_5"], [(50, "renders")], {5: 2}, 0),
        "child_wrote_nothing": (-1, True, ["iteration_5"], [(50, "renders")], {}, 0),
        "no_train_dir_at_all": (3, True, [], [], {}, 0),
        "child_failed": (-1, True, ["iteration_5"], [(50, "renders")], {}, 2),
    }
    for name, (iteration, clear_old, pcs, stale, writes, rc) in cases.items():
        root = tmp / name
        model = _model_dir(root, pcs, stale)
        subs = {str(model.resolve()): "<MODEL>", str(model): "<MODEL>", str(data.resolve()): "<DATA>", str(data): "<DATA>",
                str(engine): "<ENGINE>", str(tmp): "<TMP>", sys.executable: "<PYTHON>"}
        seen = {}

        def fake_run(cmd, _model=model, _writes=writes, _rc=rc, _seen=seen, **kw):
            _seen["argv"] = ["<PYTHON>" if c == sys.executable else c for c in cmd]
            _seen["kw"] = {"cwd": kw.get("cwd"), "pythonpath_head": kw.get("env", {}).get("PYTHONPATH", "").split(os.pathsep)[0],
                           "capture_output": kw.get("capture_output"), "text": kw.get("text"), "other": sorted(set(kw) - {"cwd", "env", "capture_output", "text"})}
            _seen["train_tree_at_launch"] = _tree(_model / "train")
            for it, n in _writes.items():
                d = _model / "train" / f"ours_{it}" / "renders"
                d.mkdir(parents=True, exist_ok=True)
                for i in range(n):
                    put_png(d / f"{i:05d}.png", np.zeros((2, 2, 3), np.uint8))
                (d / "not_a_frame.txt").write_text("x")
            return _Result(_rc, stdout="o" * 2500 + "<stdout tail>", stderr="e" * 2500 + "<stderr tail>")
        monkey(subprocess, "run", fake_run)
        rec = _call(rs.render_with_gaussians, str(model), str(data), iteration=iteration, clear_old_renders=clear_old, subs=subs)
        if "returned" in rec:
            rec["returned"] = _norm(rec["returned"], subs)
        seen["argv"] = [_norm(c, subs) for c in seen.get("argv", [])]
        if "kw" in seen:
            seen["kw"]["cwd"] = _norm(str(seen["kw"]["cwd"]), subs)
            seen["kw"]["pythonpath_head"] = _norm(seen["kw"]["pythonpath_head"], subs)
        out[name] = {**rec, "launch": seen, "train_tree_after": _tree(model / "train")}
    # the launcher refuses to start without the engine's entry point
    monkey(rs, "RENDER_SCRIPT", engine / "absent.py")
    out["no_render_script"] = _call(rs.render_with_gaussians, str(tmp / "m"), str(data), subs={str(engine): "<ENGINE>"})
    monkey(rs, "RENDER_SCRIPT", engine / "render.py")
    # relative paths are made absolute in the argv
    model = _model_dir(tmp / "rel", ["iteration_2"], [])
    seen = {}

    def fake_run_rel(cmd, **kw):
        seen["argv"] = list(cmd[2:])
        (model / "train" / "ours_2" / "renders").mkdir(parents=True)
        return _Result(0)
    monkey(subprocess, "run", fake_run_rel)
    cwd = os.getcwd()
    os.chdir(tmp / "rel")
    try:
        rec = _call(rs.render_with_gaussians, "model", "../data", subs={})
    finally:
        os.chdir(cwd)
    subs = {str((tmp / "rel").resolve()): "<CWD>", str(tmp.resolve()): "<TMP>", str(tmp / "rel"): "<CWD>", str(tmp): "<TMP>"}
    out["relative_paths"] = {"returned": rec.get("returned"), "argv": [_norm(c, subs) for c in seen["argv"]]}
    return out


def stitch(rs, tmp: Path, monkey) -> dict:
    tmp = Path(tmp)
    frames = tmp / "frames"
    frames.mkdir()
    for name in ("00010.png", "00002.png", "b.png", "00001.PNG", "notes.txt", "a.png"):
        put_png(frames / name, np.zeros((2, 2, 3), np.uint8))
    monkey(rs, "_get_ffmpeg_path", lambda: "/opt/fake/ffmpeg")
    out = {}
    for name, rc, fps, target in (("ok", 0, 30, tmp / "videos" / "deep" / "out.mp4"), ("fps_24_bare_name", 0, 24, Path("bare.mp4")),
                                  ("failed", 1, 30, tmp / "v2" / "out.mp4")):
        seen = {}

        def fake_run(cmd, _seen=seen, _rc=rc, **kw):
            staging = os.path.dirname(cmd[cmd.index("-i") + 1])
            _seen["staging_listing"] = sorted(os.listdir(staging))
            _seen["staging_prefix"] = os.path.basename(staging)[:7]
            _seen["staging"] = staging
            _seen["argv"] = [c.replace(staging, "<STAGING>") for c in cmd]
            _seen["kw"] = {k: kw[k] for k in sorted(kw)}
            return _Result(_rc, stderr="ffmpeg says no")
        monkey(subprocess, "run", fake_run)
        cwd = os.getcwd()
        os.chdir(tmp)
        try:
            rec = _call(rs.stitch_video, str(frames), str(target), fps=fps, subs={str(tmp): "<TMP>"})
        finally:
            os.chdir(cwd)
        rec.pop("returned", None)
        seen["argv"] = [_norm(c, {str(tmp): "<TMP>"}) for c in seen["argv"]]
        seen["staging_removed"] = not os.path.exists(seen.pop("staging"))
        out[name] = {**rec, "launch": seen, "parent_created": target.parent.exists() if target.is_absolute() else None}
    (tmp / "empty").mkdir()
    (tmp / "empty" / "x.txt").write_text("x")
    monkey(subprocess, "run", lambda *a, **k: (_ for _ in ()).throw(AssertionError("ffmpeg must not be launched")))
    out["no_frames"] = _call(rs.stitch_video, str(tmp / "empty"), str(tmp / "v3" / "o.mp4"), subs={str(tmp): "<TMP>"})
    out["no_frames"]["parent_created_before_refusal"] = (tmp / "v3").exists()
    return out


def deformation_map(rs, tmp: Path) -> dict:
    tmp = Path(tmp)
    subs = {str(tmp): "<TMP>"}
    (tmp / "list.json").write_text("[1, 2]")
    (tmp / "scalar.json").write_text("3")
    (tmp / "ok.json").write_text(json.dumps({"jaw_axis": 2, "lefort_scale": 1.5, "extra": [1]}))
    (tmp / "broken.json").write_text("{not json")
    out = {"none": _call(rs.load_deformation_map, None), "empty": _call(rs.load_deformation_map, ""),
           "missing": _call(rs.load_deformation_map, str(tmp / "absent.json"), subs=subs),
           "list": _call(rs.load_deformation_map, str(tmp / "list.json"), subs=subs),
           "scalar": _call(rs.load_deformation_map, str(tmp / "scalar.json"), subs=subs),
           "ok": _call(rs.load_deformation_map, str(tmp / "ok.json"), subs=subs)}
    broken = _call(rs.load_deformation_map, str(tmp / "broken.json"), subs=subs)
    out["broken_raises"] = broken["raised"][0]
    return out


# ------------------------------------------------------------------ single_frame_experiment.build_single_frame_dataset (reference :32-81)
def single_frame_dataset(build, make_fixture_dataset, tmp: Path) -> dict:
    """`build(data_conda, data_single)` builds the one-frame dataset the way the module under test does."""
    tmp = Path(tmp)
    src, dst = tmp / "data_conda", tmp / "data_single_frame"
    make_fixture_dataset(src, n_frames=5, with_masks=True)
    dst.mkdir()
    (dst / "left_over_from_last_time.txt").write_text("stale")          # an existing target is wiped first
    subs = {str(tmp): "<TMP>"}
    rec = _call(build, src, dst, subs=subs)
    rec["returned"] = _norm(str(rec["returned"]), subs)
    out = {**rec, "files": _tree(dst)}
    for name in ("transforms_train.json", "transforms_test.json", "transforms_val.json"):
        out[name] = (dst / name).read_text()
    b = np.load(dst / "flame_param.npz")
    out["batched"] = {k: {"shape": list(b[k].shape), "dtype": str(b[k].dtype),
                          "equals_frame0": bool(np.array_equal(b[k].reshape(-1), np.load(src / "flame_param" / "00000.npz")[k].reshape(-1)))}
                      for k in sorted(b.files)}
    out["copies_are_bytewise"] = {rel: (dst / rel).read_bytes() == (src / rel).read_bytes()
                                  for rel in ("images/00000_00.png", "flame_param/00000.npz", "fg_masks/00000_00.png", "canonical_flame_param.npz")}
    return out


# ------------------------------------------------------------------ the command lines: main() of render_surgery / train_ghost / validation_reporting
def _argv(argv):
    class _Ctx:
        def __enter__(self):
            self.old = sys.argv
            sys.argv = list(argv)

        def __exit__(self, *a):
            sys.argv = self.old
    return _Ctx()


def _tmpnames(text: str) -> str:
    """mkdtemp names differ from run to run: surgical_render_XXXXXXXX -> surgical_render_*"""
    import re
    text = re.sub(r"[^\s'\"]*surgical_render_[A-Za-z0-9_]+", "<MODIFIED>", text)
    return re.sub(r"[^\s'\"]*stitch_[A-Za-z0-9_]+", "<STAGING>", text)


def render_surgery_cli(rs, make_fixture_dataset, tmp: Path, monkey) -> dict:
    """main() end to end with both child processes stubbed: argument parsing and defaults, mm -> offsets, rig-mode fallback and the
    deformation map it switches on, the temporary dataset the engine is pointed at (its FLAME edits are read INSIDE the stub),
    pinned iteration, deterministic export, stitch, and the clean-up of the temporary dataset -- also when the engine fails."""
    tmp = Path(tmp)
    engine = tmp / "engine"
    engine.mkdir()
    (engine / "render.py").write_text("# placeholder\n")
    monkey(rs, "REPO_DIR", engine)
    monkey(rs, "RENDER_SCRIPT", engine / "render.py")
    monkey(rs, "_get_ffmpeg_path", lambda: "/opt/fake/ffmpeg")
    data = tmp / "data"
    make_fixture_dataset(data, n_frames=12)
    base = np.load(data / "flame_param" / "00003.npz")
    asset = tmp / "asset.npz"
    np.savez(asset, version=np.array([1]))
    dmap = tmp / "dmap.json"
    dmap.write_text(json.dumps({"translation_axis": 2, "jaw_axis": 1, "lefort_scale": 2.0, "bsso_scale": 0.5}))
    out = {}
    cases = {
        "defaults": (["--lefort_mm", "3", "--bsso_mm", "5"], 0),
        "hybrid_with_map_pinned_export": (["--lefort_mm", "-2.5", "--bsso_mm", "4", "--sensitivity", "1.5", "--rig_mode", "hybrid_full_head",
                                           "--canonical_head_asset", str(asset), "--deformation_map", str(dmap), "--iteration", "7", "--fps", "24",
                                           "--export_frames_dir", str(tmp / "export"), "--deterministic_max_frames", "3"], 0),
        "hybrid_without_asset_ignores_map": (["--lefort_mm", "1", "--bsso_mm", "1", "--rig_mode", "hybrid_full_head", "--deformation_map", str(dmap)], 0),
        "engine_fails": (["--lefort_mm", "1", "--bsso_mm", "1"], 3),
    }
    for name, (extra, rc) in cases.items():
        model = tmp / name / "model"
        (model / "point_cloud" / "iteration_30").mkdir(parents=True)
        video = tmp / name / "out" / "prediction.mp4"
        subs = {str(model.resolve()): "<MODEL>", str(model): "<MODEL>", str(data.resolve()): "<DATA>", str(data): "<DATA>", str(engine): "<ENGINE>",
                str(tmp): "<TMP>", sys.executable: "<PYTHON>"}
        seen = {"calls": []}

        def fake_run(cmd, _model=model, _rc=rc, _seen=seen, **kw):
            if cmd[0] == "/opt/fake/ffmpeg":
                _seen["calls"].append({"ffmpeg": [_tmpnames(c) for c in cmd]})
                return _Result(0)
            src = Path(cmd[cmd.index("--source_path") + 1])
            _seen["modified_dir"] = str(src)
            a = np.load(src / "flame_param" / "00003.npz")
            _seen["calls"].append({"engine": [_tmpnames(c) for c in cmd],
                                   "translation_delta": (a["translation"] - base["translation"]).round(7).tolist(),
                                   "jaw_delta": (a["jaw_pose"] - base["jaw_pose"]).round(7).tolist(),
                                   "dataset_files": sorted(p.name for p in src.iterdir())})
            if _rc == 0:
                it = cmd[cmd.index("--iteration") + 1]
                d = _model / "train" / f"ours_{it}" / "renders"
                d.mkdir(parents=True)
                for i in range(5):
                    put_png(d / f"{i:05d}.png", np.full((2, 2, 3), i, np.uint8))
            return _Result(_rc, stdout="", stderr="engine said no")
        monkey(subprocess, "run", fake_run)
        with _argv(["render_surgery.py", "--model_path", str(model), "--data_dir", str(data), "--output", str(video), *extra]):
            rec = _call(rs.main, subs=subs)
        rec.pop("returned", None)
        rec["stdout"] = _sorted_listing([_tmpnames(l) for l in rec["stdout"]])
        if "raised" in rec:
            rec["raised"][1] = _tmpnames(rec["raised"][1])
        for c in seen["calls"]:
            for k in ("engine", "ffmpeg"):
                if k in c:
                    c[k] = [_norm(x, subs) for x in c[k]]
        out[name] = {**rec, "calls": seen["calls"], "temporary_dataset_removed": not os.path.exists(seen.get("modified_dir", "/nonexistent")),
                     "exported": _tree(tmp / "export") if "export" in " ".join(extra) else None}
    with _argv(["render_surgery.py", "--bsso_mm", "1"]):
        rec = _call(lambda: _exit_code(rs.main))
    out["missing_required_argument_exit_code"] = rec.get("returned")
    return out


def _sorted_listing(lines: list) -> list:
    """The directory listing printed behind "Contents of temp_dir:" comes in os.listdir order (file-system dependent): sorted."""
    out, i = [], 0
    while i < len(lines):
        out.append(lines[i])
        if lines[i].endswith("Contents of temp_dir:"):
            j = i + 1
            while j < len(lines) and lines[j].startswith("  "):
                j += 1
            out.extend(sorted(lines[i + 1:j]))
            i = j
        else:
            i += 1
    return out


def _exit_code(fn):
    import contextlib
    try:
        with contextlib.redirect_stderr(io.StringIO()):
            fn()
    except SystemExit as e:
        return e.code
    return None


def train_ghost_cli(tg, make_fixture_dataset, tmp: Path, monkey) -> dict:
    tmp = Path(tmp)
    monkey(tg, "validate_setup", lambda: None)
    out = {}
    for name, extra, masks in (("defaults_5000", [], False), ("explicit", ["--iterations", "30000", "--resolution", "2"], True)):
        data, model = tmp / name / "data", tmp / name / "model"
        make_fixture_dataset(data, n_frames=60, with_masks=masks)
        seen = {}

        def fake_run(cmd, _seen=seen, **kw):
            _seen["argv"] = list(cmd)
            return _Result(0)
        monkey(subprocess, "run", fake_run)
        subs = {str(data.resolve()): "<DATA>", str(data): "<DATA>", str(model.resolve()): "<MODEL>", str(model): "<MODEL>",
                str(tg.REPO_DIR): "<ENGINE>", sys.executable: "<PYTHON>", str(tmp): "<TMP>"}
        with _argv(["train_ghost.py", "--data_dir", str(data), "--output_dir", str(model), *extra]):
            rec = _call(tg.main, subs=subs)
        rec.pop("returned", None)
        rec["stdout"] = [l for l in rec["stdout"] if "manifest" not in l.lower()]          # the manifest's name carries a time stamp
        out[name] = {**rec, "argv": [_norm(c, subs) for c in seen["argv"]]}
    return out


def validation_reporting_cli(vr, tmp: Path) -> dict:
    tmp = Path(tmp)
    t = make_report_tree(tmp / "tree")
    with _argv(["validation_reporting.py", "--model_path", str(t["model"]), "--deterministic_frames_dir", str(t["det"]), "--output_dir", str(tmp / "rep")]):
        rec = _call(vr.main, subs={str(tmp): "<TMP>"})
    rec.pop("returned", None)
    return {**rec, "files": _tree(tmp / "rep"), "count": json.loads((tmp / "rep" / "strict_scores.json").read_text())["summary"]["count"],
            "missing_required_argument_exit_code": _with_argv_exit(["validation_reporting.py"], vr.main)}


def _with_argv_exit(argv, fn):
    with _argv(argv):
        return _exit_code(fn)


# ------------------------------------------------------------------ flame_fitter.detect_landmarks_mediapipe (reference flame_fitter.py:45-66, 200-244)
def detect_landmarks(ff, tmp: Path) -> dict:
    """The glue around the third-party detector, with deterministic stand-ins for `cv2` and `mediapipe` registered in sys.modules
    (the detector itself is out of scope): which 68 of the mesh's landmarks are taken and in which order (MEDIAPIPE_TO_68),
    the pixel scaling, None for an unreadable image and for a frame without a face, the FaceMesh options, the printed lines."""
    import types
    tmp = Path(tmp)
    d = tmp / "frames"
    d.mkdir()
    for name in ("00000.png", "00001.png", "00002.png", "00003.png", "notes.txt", "00004.PNG"):
        (d / name).write_bytes(b"x")
    log = {"facemesh_kwargs": None, "closed": 0, "cvt_codes": []}
    cv2 = sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    mp = sys.modules.setdefault("mediapipe", types.ModuleType("mediapipe"))
    sizes = {"00000.png": (48, 64), "00001.png": (48, 64), "00003.png": (30, 40)}       # (h, w)

    def imread(path):
        name = os.path.basename(path)
        if name not in sizes:
            return None                              # 00002.png: unreadable
        h, w = sizes[name]
        img = np.zeros((h, w, 3), np.uint8)
        img[0, 0, 0] = int(name[:5])                 # the frame number travels in a pixel
        return img

    def cvt(img, code):
        log["cvt_codes"].append(code)
        return img
    cv2.imread, cv2.cvtColor, cv2.COLOR_BGR2RGB = imread, cvt, 4

    class Landmark:
        def __init__(self, i, f):
            self.x, self.y, self.z = ((i * 37 + f * 11) % 1000) / 1000.0, ((i * 91 + f * 7) % 997) / 997.0, 0.0

    class FaceMesh:
        def __init__(self, **kw):
            log["facemesh_kwargs"] = {k: kw[k] for k in sorted(kw)}

        def process(self, rgb):
            f = int(rgb[0, 0, 0])
            faces = [] if f == 1 else [types.SimpleNamespace(landmark=[Landmark(i, f) for i in range(478)])]
            return types.SimpleNamespace(multi_face_landmarks=faces)

        def close(self):
            log["closed"] += 1
    mp.solutions = types.SimpleNamespace(face_mesh=types.SimpleNamespace(FaceMesh=FaceMesh))
    rec = _call(ff.detect_landmarks_mediapipe, str(d))
    found = rec.pop("returned")
    return {**rec, "table": [int(i) for i in ff.MEDIAPIPE_TO_68], "log": log,
            "landmarks": [None if l is None else {"dtype": str(l.dtype), "shape": list(l.shape), "values": np.asarray(l, np.float64).round(6).tolist()} for l in found]}


# ------------------------------------------------------------------ flame_fitter.fit_video + its command line (reference flame_fitter.py:447-490)
def fit_video(ff, tmp: Path, monkey, pkl: str, lmk_npy: str, device: str, lmk2d, size_wh) -> dict:
    """detect -> fit -> np.savez with the detector replaced by `lmk2d` (list of (68,2) arrays / None) and 3 fit iterations,
    through the command line (`main()`), on `device`; the size of the first frame is read from a real PNG of `size_wh`
    (the reference reads it with cv2.imread: the stand-in returns an array of that size).  Returns the printed lines, the saved
    arrays (the caller compares them with the tolerance of the fit) and the refusal without the FLAME pickle."""
    import types
    tmp = Path(tmp)
    d = tmp / "frames"
    d.mkdir()
    w, h = size_wh
    for i in range(len(lmk2d)):
        put_png(d / f"{i:05d}.png", np.zeros((h, w, 3), np.uint8))
    cv2 = sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    cv2.imread = lambda path: np.zeros((h, w, 3), np.uint8)
    monkey(ff, "detect_landmarks_mediapipe", lambda images_dir: [None if l is None else np.array(l, np.float32) for l in lmk2d])
    monkey(ff, "FLAME_LMK_PATH", Path(lmk_npy))
    subs = {str(tmp): "<TMP>"}
    monkey(ff, "FLAME_MODEL_PATH", tmp / "absent.pkl")
    out = {"no_model": _call(ff.fit_video, str(d), str(tmp / "x.npz"), device, 3, subs=subs)}
    monkey(ff, "FLAME_MODEL_PATH", Path(pkl))
    with _argv(["flame_fitter.py", "--images_dir", str(d), "--output", str(tmp / "fit.npz"), "--device", device, "--n_iters", "3"]):
        rec = _call(ff.main, subs=subs)
    rec.pop("returned", None)
    rec["stdout"] = [l for l in rec["stdout"] if l.strip()]
    res = np.load(tmp / "fit.npz")
    out["cli"] = {**rec, "keys": sorted(res.files), "shapes": {k: list(res[k].shape) for k in sorted(res.files)},
                  "dtypes": {k: str(res[k].dtype) for k in sorted(res.files)}}
    out["_arrays"] = {k: res[k] for k in res.files if k not in ("static_offset", "dynamic_offset")}      # not JSON: stripped by the caller
    out["missing_required_argument_exit_code"] = _with_argv_exit(["flame_fitter.py", "--output", "x"], ff.main)
    return out


# ------------------------------------------------------------------ the smaller helpers of train_ghost and render_surgery
def helper_functions(rs, tg, make_fixture_dataset, tmp: Path, monkey) -> dict:
    """train_ghost.validate_setup / validate_data (exact messages), _collect_checkpoint_lineage, write_experiment_manifest (whole
    payload minus its time stamps); render_surgery._get_ffmpeg_path (bundled, system, absent) and export_deterministic_frames with
    an index file (list form, {"indices": ...} form, out-of-range indices dropped, a non-integer refused, no frames refused)."""
    import re
    import types
    tmp = Path(tmp)
    subs = {str(tmp.resolve()): "<TMP>", str(tmp): "<TMP>"}
    out = {}
    # ---- validate_setup
    monkey(tg, "REPO_DIR", tmp / "no_repo")
    monkey(tg, "TRAIN_SCRIPT", tmp / "no_repo" / "train.py")
    out["setup_no_repo"] = _call(tg.validate_setup, subs=subs)
    (tmp / "repo").mkdir()
    monkey(tg, "REPO_DIR", tmp / "repo")
    monkey(tg, "TRAIN_SCRIPT", tmp / "repo" / "train.py")
    out["setup_no_script"] = _call(tg.validate_setup, subs=subs)
    (tmp / "repo" / "train.py").write_text("#\n")
    out["setup_ok"] = _call(tg.validate_setup, subs=subs)
    # ---- validate_data: each refusal in the order the function checks
    d = tmp / "vd"
    d.mkdir()
    steps = []
    steps.append(_call(tg.validate_data, str(d), subs=subs))
    (d / "transforms_train.json").write_text("{}")
    steps.append(_call(tg.validate_data, str(d), subs=subs))
    (d / "transforms_test.json").write_text("{}")
    steps.append(_call(tg.validate_data, str(d), subs=subs))
    (d / "flame_param.npz").write_bytes(b"x")
    steps.append(_call(tg.validate_data, str(d), subs=subs))
    (d / "images").mkdir()
    (d / "images" / "a.jpg").write_bytes(b"x")
    steps.append(_call(tg.validate_data, str(d), subs=subs))
    for i in range(3):
        put_png(d / "images" / f"{i:05d}_00.png", np.zeros((2, 2, 3), np.uint8))
    steps.append(_call(tg.validate_data, str(d), subs=subs))
    out["validate_data_steps"] = steps
    # ---- checkpoint lineage + manifest
    m = tmp / "model"
    out["lineage_missing_dir"] = tg._collect_checkpoint_lineage(str(m))
    m.mkdir()
    for name, size in (("chkpnt500.pth", 7), ("chkpnt10000.pth", 11), ("chkpnt_note.txt", 3), ("other.pth", 5)):
        (m / name).write_bytes(b"z" * size)
        os.utime(m / name, (1700000000, 1700000000 + size))
    out["lineage"] = tg._collect_checkpoint_lineage(str(m))
    data = tmp / "data"
    make_fixture_dataset(data, n_frames=6)
    rec = _call(tg.write_experiment_manifest, str(data), str(m), 1234, 2, ["python", "train.py", "--x"], {"note": "n", "k": [1, 2]}, subs=subs)
    path = Path(str(rec.pop("returned")))
    payload = json.loads(path.read_text())
    out["manifest"] = {"stdout": [re.sub(r"\d{8}T\d{6}Z", "<STAMP>", l) for l in rec["stdout"]],
                       "name_is_utc_stamp": bool(re.fullmatch(r"\d{8}T\d{6}Z\.json", path.name)), "parent": _norm(str(path.parent), subs),
                       "keys_in_order": list(payload.keys()), "created_utc_is_iso_utc": payload["created_utc"].endswith("+00:00"),
                       "payload": {k: (_norm(v, subs) if isinstance(v, str) else v) for k, v in payload.items() if k not in ("created_utc", "dataset_fingerprint")},
                       "fingerprint_keys": sorted(payload["dataset_fingerprint"].keys())}
    # ---- _get_ffmpeg_path
    fake = types.ModuleType("imageio_ffmpeg")
    fake.get_ffmpeg_exe = lambda: "/bundled/ffmpeg"
    saved = sys.modules.get("imageio_ffmpeg", "absent")
    try:
        sys.modules["imageio_ffmpeg"] = fake
        ff = {"bundled": _call(rs._get_ffmpeg_path)}
        sys.modules["imageio_ffmpeg"] = None                      # import raises ImportError
        monkey(rs.shutil, "which", lambda name: "/usr/bin/" + name)
        ff["system"] = _call(rs._get_ffmpeg_path)
        monkey(rs.shutil, "which", lambda name: None)
        ff["absent"] = _call(rs._get_ffmpeg_path)
    finally:
        if saved == "absent":
            sys.modules.pop("imageio_ffmpeg", None)
        else:
            sys.modules["imageio_ffmpeg"] = saved
    out["ffmpeg_path"] = ff
    # ---- export_deterministic_frames with an index file
    frames = tmp / "frames"
    frames.mkdir()
    for i in range(7):
        put_png(frames / f"{i:05d}.png", np.full((2, 2, 3), i, np.uint8))
    (frames / "skip.txt").write_text("x")
    ex = {}
    for name, content in (("list", [5, 0, 99, 3, -1, 3]), ("dict", {"indices": [6, 2], "comment": "x"}), ("bad_entry", [1, "2"]), ("bad_type", {"indices": "0,1"}),
                          ("dict_without_key", {"other": 1})):
        idx = tmp / f"idx_{name}.json"
        idx.write_text(json.dumps(content))
        rec = _call(rs.export_deterministic_frames, str(frames), str(tmp / f"exp_{name}"), str(idx), 24, subs=subs)
        if "returned" in rec:
            rec["returned"] = _norm(rec["returned"], subs)
            man = json.loads((tmp / f"exp_{name}" / "deterministic_indices_manifest.json").read_text())
            rec["manifest"] = {**man, "source_frames_dir": _norm(man["source_frames_dir"], subs)}
            rec["files"] = _tree(tmp / f"exp_{name}")
        ex[name] = rec
    (tmp / "noframes").mkdir()
    ex["no_frames"] = _call(rs.export_deterministic_frames, str(tmp / "noframes"), str(tmp / "exp_none"), None, 24, subs=subs)
    out["export_with_index_file"] = ex
    return out

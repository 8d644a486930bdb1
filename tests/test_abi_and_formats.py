"""CPU: the C-ABI library loads and exports every symbol include/omfs_splat.h declares (no compute
calls: there is no GPU here); the ctypes table covers exactly those symbols; file formats round-trip."""
import ctypes
import json
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]


def declared_symbols():
    text = (ROOT / "include" / "omfs_splat.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(omfs_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from omfs_4d_video_gen_amd import _lib as L
    lib = L.load()
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/omfs_splat.h but not exported"
    assert sorted(L.SIGNATURES) == syms, "ctypes table and header disagree"
    assert lib.omfs_abi_version() == L.ABI_VERSION == int(re.search(r"#define OMFS_ABI_VERSION (\d+)", (ROOT / "include" / "omfs_splat.h").read_text()).group(1))


def test_argument_validation_fails_loudly_without_touching_the_gpu():
    from omfs_4d_video_gen_amd import _lib as L
    lib = L.load()
    rc = lib.omfs_adam_step(0, 0, 0, 0, 0, 0, ctypes.byref(L.AdamParamsC()), 0)
    assert rc == -1 and b"null pointer" in lib.omfs_last_error()
    with pytest.raises(L.OmfsError, match="omfs_adam_step failed"):
        L.check(rc, "omfs_adam_step")


def test_struct_layouts_match_the_header():
    """sizeof of every ctypes struct equals what the C compiler lays out (guards silent ABI drift)."""
    import subprocess
    import tempfile
    from omfs_4d_video_gen_amd import _lib as L
    names = {"omfs_flame_rig": L.FlameRigC, "omfs_simpleflame": L.SimpleFlameC, "omfs_camera": L.CameraC, "omfs_gaussians": L.GaussiansC,
             "omfs_raster_buffers": L.RasterBuffersC, "omfs_grad_buffers": L.GradBuffersC, "omfs_reg_params": L.RegParamsC,
             "omfs_adam_params": L.AdamParamsC, "omfs_view_set": L.ViewSetC, "omfs_view_step": L.ViewStepC,
             "omfs_lr_schedule": L.LrScheduleC, "omfs_flame_fit": L.FlameFitC, "omfs_densify_params": L.DensifyParamsC,
             "omfs_step_state": L.StepStateC}
    # every struct the header declares has a ctypes mirror here: a struct added to the ABI must be added to this table
    import re
    declared = set(re.findall(r"^}\s*(omfs_\w+);", (ROOT / "include" / "omfs_splat.h").read_text(), flags=re.M))
    assert declared == set(names), declared ^ set(names)
    src = '#include <stdio.h>\n#include "omfs_splat.h"\nint main(){' + "".join(f'printf("{n} %zu\\n", sizeof({n}));' for n in names) + "return 0;}"
    with tempfile.TemporaryDirectory() as d:
        (Path(d) / "s.c").write_text(src)
        subprocess.check_call(["gcc", "-I", str(ROOT / "include"), "-o", f"{d}/s", f"{d}/s.c"])
        out = subprocess.check_output([f"{d}/s"], text=True)
    for line in out.strip().splitlines():
        n, size = line.split()
        assert ctypes.sizeof(names[n]) == int(size), n


def test_png_and_ply_roundtrip(tmp_path):
    from omfs_4d_video_gen_amd.engine import io_formats as IO
    from omfs_4d_video_gen_amd.engine import synthetic
    rng = np.random.default_rng(1)
    for shape in ((7, 5, 3), (16, 16, 4), (3, 9)):
        img = rng.integers(0, 255, shape).astype(np.uint8)
        IO.write_png(tmp_path / "a.png", img)
        back = IO.read_png(tmp_path / "a.png")
        assert np.array_equal(back if img.ndim == 3 else back[:, :, 0], img)
    with pytest.raises(ValueError):
        (tmp_path / "bad.png").write_bytes(b"not a png")
        IO.read_png(tmp_path / "bad.png")
    # ready-made scanlines (what the GPU hands over: filter byte 0 + RGB per row) encode to the same picture
    img = rng.integers(0, 255, (11, 13, 3)).astype(np.uint8)
    rows = np.zeros((11, 1 + 3 * 13), np.uint8)
    rows[:, 1:] = img.reshape(11, -1)
    (tmp_path / "rows.png").write_bytes(IO.encode_png_rows(rows, 13, 11))
    assert np.array_equal(IO.read_png(tmp_path / "rows.png"), img)
    with pytest.raises(ValueError):
        IO.encode_png_rows(rows[:, 1:], 13, 11)
    g = synthetic.make_gaussians(257, 100, 3)
    IO.save_gaussian_ply(tmp_path / "pc" / "point_cloud.ply", g)
    back = IO.load_gaussian_ply(tmp_path / "pc" / "point_cloud.ply")
    for k in g:
        assert np.array_equal(back[k], g[k]), k


def test_dataset_write_load_roundtrip(tmp_path):
    from omfs_4d_video_gen_amd.engine import io_formats as IO
    from omfs_4d_video_gen_amd.engine import synthetic
    from omfs_4d_video_gen_amd.train_ghost import validate_data, run_quality_gates
    T = 60
    seq = synthetic.make_flame_sequence(T, 4)
    cams = [synthetic.make_camera(32, 24, yaw=0.01 * i) for i in range(T)]
    imgs = [np.full((24, 32, 3), i, np.uint8) for i in range(T)]
    IO.write_dataset(tmp_path / "ds", cams, list(range(T)), imgs, seq, fg_masks=True)
    validate_data(str(tmp_path / "ds"))
    run_quality_gates(str(tmp_path / "ds"))
    sp = IO.load_split(str(tmp_path / "ds"), "train")
    assert len(sp["frames"]) == 54 and sp["timestep_of_frame"] == list(range(54))
    for k in ("expr", "rotation", "jaw_pose", "translation", "neck_pose", "eyes_pose"):
        assert np.array_equal(sp["flame"][k], seq[k][:54]), k
    assert np.array_equal(sp["flame"]["shape"], seq["shape"]) and sp["flame"]["static_offset"].shape == (1, 5143, 3)
    cam = IO.camera_from_frame(sp["frames"][7], sp["top"])
    assert np.allclose(cam["world_to_view"], cams[7]["world_to_view"], atol=1e-6) and cam["width"] == 32
    assert cam["fl_x"] == pytest.approx(cams[7]["fl_x"], rel=1e-6)
    test = IO.load_split(str(tmp_path / "ds"), "test")
    assert len(test["frames"]) == 6 and np.array_equal(test["flame"]["expr"], seq["expr"][54:])
    # the default VHAP transform (preprocess_video.py:370) is a camera one unit in front of the head
    c = IO.camera_from_frame({"transform_matrix": [[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 1], [0, 0, 0, 1]], "w": 8, "h": 8, "camera_angle_x": 0.5}, {})
    assert np.allclose(c["cam_pos"], [0, 0, 1]) and np.allclose(c["world_to_view"][:3, :3], np.diag([1, -1, -1]))


def test_flame_pickle_with_chumpy_objects_loads_without_chumpy(tmp_path):
    """The released FLAME pickles hold chumpy arrays; the loader must not need the chumpy package."""
    import pickle
    import sys
    import types
    from omfs_4d_video_gen_amd.engine.flame_rig import FlameRig, load_flame_pickle
    from omfs_4d_video_gen_amd.engine import synthetic
    rig = synthetic.make_rig(0)
    fake = types.ModuleType("chumpy")
    fake_ch = types.ModuleType("chumpy.ch")

    class Ch:                                   # pickled by reference to chumpy.ch.Ch, state = {"x": ndarray, ...}
        def __init__(self, x):
            self.x = np.asarray(x)
            self._dirty_vars = set()
    Ch.__module__, Ch.__qualname__ = "chumpy.ch", "Ch"
    fake_ch.Ch = Ch
    fake.ch = fake_ch
    sys.modules["chumpy"], sys.modules["chumpy.ch"] = fake, fake_ch
    try:
        import scipy.sparse as sp
        kt = np.stack([np.array([2 ** 32 - 1, 0, 1, 1, 1], np.int64), np.arange(5)])
        blob = {"v_template": Ch(rig.v_template), "shapedirs": Ch(rig.shapedirs), "posedirs": Ch(rig.posedirs),
                "J_regressor": sp.csc_matrix(rig.J_regressor), "weights": Ch(rig.weights), "f": rig.faces.astype(np.uint32),
                "kintree_table": kt}
        path = tmp_path / "flame_like.pkl"
        with open(path, "wb") as f:
            pickle.dump(blob, f, protocol=2)
    finally:
        del sys.modules["chumpy"], sys.modules["chumpy.ch"]
    m = load_flame_pickle(str(path))
    assert type(m["v_template"]).__name__ == "_ChumpyArray"
    assert np.array_equal(np.asarray(m["v_template"], np.float32), rig.v_template)
    r = FlameRig.from_pickle(str(path))
    assert r.n_verts == rig.v_template.shape[0] and r.n_faces == rig.faces.shape[0]
    assert np.array_equal(r.shapedirs, rig.shapedirs.astype(np.float32)) and np.array_equal(r.faces, rig.faces.astype(np.int32))


def test_rgba_images_return_their_matte(tmp_path):
    from omfs_4d_video_gen_amd.engine import io_formats as IO
    rng = np.random.default_rng(1)
    rgba = rng.integers(0, 256, (9, 13, 4), dtype=np.uint8)
    IO.write_png(tmp_path / "a.png", rgba)
    IO.write_png(tmp_path / "b.png", rgba[:, :, :3])
    rgb, alpha = IO.load_image_rgba(tmp_path / "a.png")
    assert np.array_equal(rgb, rgba[:, :, :3]) and np.array_equal(alpha, rgba[:, :, 3])
    rgb2, alpha2 = IO.load_image_rgba(tmp_path / "b.png")
    assert np.array_equal(rgb2, rgba[:, :, :3]) and alpha2 is None
    assert np.array_equal(IO.load_image_rgb(tmp_path / "a.png"), rgba[:, :, :3])

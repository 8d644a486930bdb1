"""The 5143-vertex head: FLAME's 5023 vertices + 120 procedural teeth vertices (engine/flame_rig.add_teeth), the vertex
count of `static_offset (1,5143,3)` / `dynamic_offset (T,5143,3)` in the reference's npz contract
(`02_Visual_Engine/flame_fitter.py:439-440`, `preprocess_video.py:404-416`).  CPU: construction, rigging (checked through
the oracle's full FLAME forward) and the pickle path."""
import numpy as np
import torch

import helpers as H
from omfs_4d_video_gen_amd.engine import synthetic
from omfs_4d_video_gen_amd.engine.flame_rig import FlameRig, N_TEETH_FACES, N_TEETH_VERTS, pose_rotmats


def test_rig_has_flames_vertices_plus_teeth(rig_small):
    r = rig_small
    assert r.n_base_verts == 5023 and N_TEETH_VERTS == 120 and r.v_template.shape == (5143, 3)
    assert r.faces.shape[0] == r.n_base_faces + N_TEETH_FACES == 9951 + 168
    assert r.faces[:r.n_base_faces].max() < 5023                     # FLAME's own faces do not reference teeth
    tf = r.faces[r.n_base_faces:]
    assert tf.min() >= 5023 and set(np.unique(tf)) == set(range(5023, 5143))
    v = r.v_template
    area = np.linalg.norm(np.cross(v[tf[:, 1]] - v[tf[:, 0]], v[tf[:, 2]] - v[tf[:, 0]]), axis=1)
    assert area.min() > 1e-6                                          # no degenerate teeth triangle (frames are well defined)
    up, low = v[5023:5023 + 15], v[5023 + 15:5023 + 30]               # upper / lower roots
    assert (up[:, 1] > low[:, 1]).all()
    lips_z = np.minimum(v[r.lip_upper][:, 2], v[r.lip_lower][:, 2])
    assert (v[5023:, 2].reshape(8, 15) < lips_z[None, :]).all()       # every column sits behind its lip vertices
    # rigging: one-hot on the neck joint (upper arch) / jaw joint (lower arch), outside the joint regressor,
    # identity blendshapes of the lip vertex, no expression or pose-corrective displacement
    w = r.weights[5023:].reshape(8, 15, 5)
    upper_rows, lower_rows = [0, 2, 4, 5], [1, 3, 6, 7]
    assert (w[upper_rows][..., 1] == 1).all() and (w[lower_rows][..., 2] == 1).all() and np.allclose(w.sum(-1), 1)
    assert not r.J_regressor[:, 5023:].any() and not r.posedirs[5023:].any() and not r.shapedirs[5023:, :, 300:].any()
    assert np.array_equal(r.shapedirs[5023:5038, :, :300], r.shapedirs[r.lip_upper][:, :, :300])
    assert np.array_equal(r.shapedirs[5038:5053, :, :300], r.shapedirs[r.lip_lower][:, :, :300])


def test_lower_teeth_follow_the_jaw_and_upper_teeth_the_head(rig_small):
    from oracle import torch_splat as O
    rig = H.oracle_rig(rig_small)
    seq = synthetic.make_flame_sequence(1, 0, identity=True)
    shape, expr = torch.zeros(300), torch.zeros(1, 100)
    I = torch.eye(3).expand(1, 5, 3, 3).clone()
    rest = O.flame_lbs(rig, shape, expr, I, torch.zeros(1, 3))[0]
    jaw = I.clone()
    jaw[0, 2] = O.rodrigues(torch.tensor([[0.3, 0.0, 0.0]]))[0]
    opened = O.flame_lbs(rig, shape, expr, jaw, torch.zeros(1, 3))[0]
    teeth = slice(5023, 5143)
    upper = np.r_[5023:5038, 5053:5068, 5083:5113]
    lower = np.r_[5038:5053, 5068:5083, 5113:5143]
    assert torch.allclose(opened[upper], rest[upper], atol=1e-7)                # the jaw does not move the upper arch
    J = torch.einsum("jv,vc->jc", rig["J_regressor"], rest)                      # joints at rest (teeth carry no weight)
    want = (rest[lower] - J[2]) @ jaw[0, 2].T + J[2]                             # rigid rotation about the jaw joint
    assert torch.allclose(opened[lower], want, atol=1e-6) and (opened[lower] - rest[lower]).abs().max() > 1e-3
    # expressions leave the teeth alone; the head rotation carries both arches
    e = torch.zeros(1, 100); e[0, :10] = 1.0
    assert torch.allclose(O.flame_lbs(rig, shape, e, I, torch.zeros(1, 3))[0][teeth], rest[teeth], atol=1e-7)
    neck = I.clone()
    neck[0, 1] = O.rodrigues(torch.tensor([[0.0, 0.4, 0.0]]))[0]
    turned = O.flame_lbs(rig, shape, expr, neck, torch.zeros(1, 3))[0]
    want = (rest[teeth] - J[1]) @ neck[0, 1].T + J[1]
    assert torch.allclose(turned[teeth], want, atol=1e-6)
    assert seq["static_offset"].shape == (1, 5143, 3) and seq["dynamic_offset"].shape == (1, 5143, 3)


def test_pickle_holds_flame_only_and_the_loader_appends_the_teeth(rig_small, tmp_path):
    from omfs_4d_video_gen_amd.engine.flame_rig import load_flame_pickle
    pkl = tmp_path / "flame2023.pkl"
    synthetic.write_flame_pickle(rig_small, str(pkl))
    raw = load_flame_pickle(str(pkl))
    assert np.asarray(raw["v_template"]).shape == (5023, 3) and np.asarray(raw["f"]).shape == (9951, 3)
    bare = FlameRig.from_pickle(str(pkl))
    assert bare.n_verts == 5023                                       # no lip rings known: FLAME as it is
    full = FlameRig.from_pickle(str(pkl), lip_rings=(rig_small.lip_upper, rig_small.lip_lower))
    ref = FlameRig.from_synthetic(rig_small)
    for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "weights", "faces"):
        assert np.array_equal(getattr(full, k), getattr(ref, k)), k
    # lip rings from a FLAME_masks.pkl next to the model (`lips` region): 15 + 15 columns, teeth appended
    import pickle
    lips = np.concatenate([rig_small.lip_upper, rig_small.lip_lower])
    with open(tmp_path / "FLAME_masks.pkl", "wb") as f:
        pickle.dump({"lips": lips}, f, protocol=2)
    auto = FlameRig.from_pickle(str(pkl))
    assert auto.n_verts == 5143 and auto.n_faces == 9951 + 168 and np.array_equal(auto.v_template, ref.v_template)
    assert pose_rotmats(synthetic.make_flame_sequence(2, 0)).shape == (2, 5, 3, 3)

"""ORACLE (test infrastructure only): numpy restatement of the adaptive density control step the engine runs in HIP
(omfs_densify_classify / _scan / _compact; SURVEY.md Appendix A item 10 -- the absent upstream train.py's
densify_and_prune; call site 02_Visual_Engine/train_ghost.py:227-271).  Parity unpinned: upstream is not in the
reference, the rule set and the split sampler are this build's frozen conventions.

Integer work (classification bits away from the thresholds, output positions, parent triangles, the generator's
32-bit words) is bit-exact; the samples go through log/cos and are compared to 1e-5."""
import numpy as np

P_XYZ, P_SCALE, P_ROT, P_OPACITY, NPLANES = 0, 3, 6, 10, 59


def mix32(x):
    x = np.asarray(x, np.uint32).copy()
    x ^= x >> np.uint32(16); x *= np.uint32(0x7feb352d); x ^= x >> np.uint32(15); x *= np.uint32(0x846ca68b); x ^= x >> np.uint32(16)
    return x


def normal_samples(seed_lo, seed_hi, ids, slot):
    """One standard normal per id for generator slot `slot` (child * 3 + axis)."""
    with np.errstate(over="ignore"):
        k = mix32(np.uint32(seed_lo) ^ mix32(np.uint32(seed_hi) + np.uint32(0x9e3779b9)))
        a = mix32(k ^ mix32(np.asarray(ids, np.uint32) * np.uint32(6) + np.uint32(slot)))
        b = mix32(a + np.uint32(0x85ebca6b))
    u1 = ((a >> np.uint32(8)).astype(np.float32) + np.float32(1.0)) * np.float32(1.0 / 16777216.0)
    u2 = (b >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return (np.sqrt(-2.0 * np.log(u1.astype(np.float64))) * np.cos(6.283185307179586 * u2.astype(np.float64))).astype(np.float32)


def classify(params, binding, face_scale, stats, grad_threshold, size_threshold, min_opacity, prune_size):
    """-> (cls uint8 [n], margin [n]): margin = smallest relative distance of a tested quantity to its threshold."""
    p = params.astype(np.float64)
    grad = stats[0].astype(np.float32) / np.maximum(stats[1].astype(np.float32), np.float32(1.0))
    world_max = np.exp(p[P_SCALE:P_SCALE + 3].max(0)) * face_scale[binding].astype(np.float64)
    hot = grad >= np.float32(grad_threshold)
    small = world_max <= size_threshold
    split, clone = hot & ~small, hot & small
    opacity = 1.0 / (1.0 + np.exp(-p[P_OPACITY]))
    prune = split | (opacity < min_opacity)
    if prune_size > 0:
        prune |= world_max > prune_size
    cls = (~prune).astype(np.uint8) | (clone.astype(np.uint8) << 1) | (split.astype(np.uint8) << 2)
    rel = lambda v, t: np.abs(v - t) / max(abs(t), 1e-30)
    margin = np.minimum(rel(world_max, size_threshold), rel(opacity, min_opacity))
    if prune_size > 0:
        margin = np.minimum(margin, rel(world_max, prune_size))
    return cls, margin


def compact(params, binding, adam_m, adam_v, cls, seed_lo, seed_hi):
    """-> (params [59][n_out], binding [n_out], m, v): [kept | clones | first children | second children]."""
    keep, clone, split = (cls & 1) > 0, (cls & 2) > 0, (cls & 4) > 0
    sp = params[:, split].astype(np.float32)
    ids = np.nonzero(split)[0]
    q = sp[P_ROT:P_ROT + 4].astype(np.float64)
    q = q / np.sqrt((q * q).sum(0, keepdims=True))
    w, x, y, z = q
    R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                  2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                  2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], 0).reshape(3, 3, -1)
    stds = np.exp(sp[P_SCALE:P_SCALE + 3].astype(np.float64))
    children = []
    for child in range(2):
        d = np.stack([normal_samples(seed_lo, seed_hi, ids, child * 3 + a).astype(np.float64) for a in range(3)], 0) * stds
        c = sp.copy()
        c[P_XYZ:P_XYZ + 3] = (sp[P_XYZ:P_XYZ + 3].astype(np.float64) + np.einsum("ijn,jn->in", R, d)).astype(np.float32)
        c[P_SCALE:P_SCALE + 3] = sp[P_SCALE:P_SCALE + 3] - np.float32(0.4700036292457356)
        children.append(c)
    new_p = np.concatenate([params[:, keep], params[:, clone]] + children, 1)
    new_b = np.concatenate([binding[keep], binding[clone], binding[split], binding[split]])
    fresh = np.zeros((NPLANES, int(clone.sum()) + 2 * int(split.sum())), np.float32)
    return new_p, new_b, np.concatenate([adam_m[:, keep], fresh], 1), np.concatenate([adam_v[:, keep], fresh], 1)

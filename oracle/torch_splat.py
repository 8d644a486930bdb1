"""ORACLE (test infrastructure, not product code): PyTorch-CPU fp32 restatement of the
FLAME-rigged Gaussian-avatar hot path.  Only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import this file.

PARITY UNPINNED for everything downstream of the FLAME parameters: the reference only
launches an un-vendored trainer/renderer (`02_Visual_Engine/train_ghost.py:227-271`,
`02_Visual_Engine/render_surgery.py:289-315` -> `gaussian_avatars_repo/{train,render}.py`,
git-ignored at `.gitignore:27`, no version pinned) and holds no test or golden vector for
it.  The algorithm restated here is the published one (3D Gaussian Splatting, Kerbl et al.
2023; GaussianAvatars, Qian et al. 2024) with every constant frozen as listed in
DESIGN.md §"Frozen conventions".  This is BASELINE.json's "PyTorch-CPU splat".

What IS pinned: `rodrigues()` below follows `SimpleFLAME._axis_angle_to_matrix`
(`02_Visual_Engine/flame_fitter.py:122-152`) and is checked against golden vectors generated
from the reference itself (tests/golden/flame_fitter_golden.npz).

All functions are differentiable where the algorithm is (autograd = backward oracle).
"""
from __future__ import annotations

import math

import torch

TILE = 16
SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
         -0.4570457994644658, 1.445305721320277, -0.5900435899266435]
PARENTS = [-1, 0, 1, 1, 1]


# --------------------------------------------------------------------------- FLAME
def rodrigues(aa: torch.Tensor) -> torch.Tensor:
    """(B,3) axis-angle -> (B,3,3).  Follows flame_fitter.py:122-152:
    angle = ||aa||, axis = aa/(angle+1e-8), R = I + sin*K + (1-cos)*K@K."""
    angle = torch.norm(aa, dim=1, keepdim=True)
    axis = aa / (angle + 1e-8)
    c = torch.cos(angle).unsqueeze(-1)
    s = torch.sin(angle).unsqueeze(-1)
    z = torch.zeros_like(axis[:, 0])
    K = torch.stack([
        torch.stack([z, -axis[:, 2], axis[:, 1]], 1),
        torch.stack([axis[:, 2], z, -axis[:, 0]], 1),
        torch.stack([-axis[:, 1], axis[:, 0], z], 1)], 1)
    I = torch.eye(3, dtype=aa.dtype).unsqueeze(0)
    return I + s * K + (1 - c) * torch.bmm(K, K)


def flame_lbs(rig: dict, shape, expr, rotmats, translation, static_offset=None, dynamic_offset=None):
    """Full FLAME forward (blendshapes + pose correctives + LBS) for B frames.

    rig: v_template (V,3), shapedirs (V,3,400), posedirs (V,3,36), J_regressor (5,V), weights (V,5).
    shape (300,), expr (B,100), rotmats (B,5,3,3) [global, neck, jaw, eyeL, eyeR], translation (B,3).
    Returns verts (B,V,3).  [NOT IN REFERENCE: SURVEY Appendix A item 1]"""
    B = expr.shape[0]
    v_t = rig["v_template"]
    sd = rig["shapedirs"]
    v_static = v_t + torch.einsum("vck,k->vc", sd[:, :, :300], shape)
    if static_offset is not None:
        v_static = v_static + static_offset.reshape(-1, 3)
    v_shaped = v_static.unsqueeze(0) + torch.einsum("vck,bk->bvc", sd[:, :, 300:300 + expr.shape[1]], expr)
    J = torch.einsum("jv,bvc->bjc", rig["J_regressor"], v_shaped)                      # (B,5,3)
    I = torch.eye(3, dtype=v_t.dtype)
    pose_feat = (rotmats[:, 1:] - I).reshape(B, 36)                                    # row-major (j,r,c)
    v_posed = v_shaped + torch.einsum("vck,bk->bvc", rig["posedirs"], pose_feat)
    # kinematic chain
    Rw = [rotmats[:, 0]]
    tw = [J[:, 0]]
    for j in range(1, 5):
        p = PARENTS[j]
        Rw.append(torch.bmm(Rw[p], rotmats[:, j]))
        tw.append(torch.bmm(Rw[p], (J[:, j] - J[:, p]).unsqueeze(-1)).squeeze(-1) + tw[p])
    A_R = torch.stack(Rw, 1)                                                            # (B,5,3,3)
    A_t = torch.stack([tw[j] - torch.bmm(Rw[j], J[:, j].unsqueeze(-1)).squeeze(-1) for j in range(5)], 1)
    W = rig["weights"]                                                                  # (V,5)
    R_blend = torch.einsum("vj,bjrc->bvrc", W, A_R)
    t_blend = torch.einsum("vj,bjc->bvc", W, A_t)
    v = torch.einsum("bvrc,bvc->bvr", R_blend, v_posed) + t_blend
    if dynamic_offset is not None:
        v = v + dynamic_offset
    return v + translation.unsqueeze(1)


def safe_normalize(x, eps=1e-20):
    return x / torch.sqrt(torch.clamp((x * x).sum(-1, keepdim=True), min=eps))


def face_frames(verts, faces):
    """verts (V,3), faces (F,3) -> R (F,3,3) columns [a0, n, a2], center (F,3), scale (F,).
    a0 = normalize(v1-v0); n = normalize(a0 x (v2-v0)); a2 = -normalize(n x a0);
    scale = (|v1-v0| + |a2.(v2-v0)|)/2.   [NOT IN REFERENCE: SURVEY Appendix A item 2]"""
    v0, v1, v2 = verts[faces[:, 0]], verts[faces[:, 1]], verts[faces[:, 2]]
    e01, e02 = v1 - v0, v2 - v0
    a0 = safe_normalize(e01)
    n = safe_normalize(torch.cross(a0, e02, dim=-1))
    a2 = -safe_normalize(torch.cross(n, a0, dim=-1))
    R = torch.stack([a0, n, a2], dim=-1)
    s0 = torch.sqrt((e01 * e01).sum(-1))
    s1 = torch.abs((a2 * e02).sum(-1))
    return R, (v0 + v1 + v2) / 3.0, (s0 + s1) * 0.5


def quat_to_rotmat(q):
    """q (N,4) = (w,x,y,z), normalised inside."""
    q = q / torch.sqrt((q * q).sum(-1, keepdim=True))
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    return torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1).reshape(-1, 3, 3)


# --------------------------------------------------------------------------- per-Gaussian
def eval_sh(sh, dirs, degree):
    """sh (N,16,3), dirs (N,3) unit.  Returns colour (N,3) = max(SH + 0.5, 0) and clamp mask."""
    res = SH_C0 * sh[:, 0]
    if degree > 0:
        x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
        res = res - SH_C1 * y * sh[:, 1] + SH_C1 * z * sh[:, 2] - SH_C1 * x * sh[:, 3]
        if degree > 1:
            xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
            res = (res + SH_C2[0] * xy * sh[:, 4] + SH_C2[1] * yz * sh[:, 5]
                   + SH_C2[2] * (2 * zz - xx - yy) * sh[:, 6] + SH_C2[3] * xz * sh[:, 7]
                   + SH_C2[4] * (xx - yy) * sh[:, 8])
            if degree > 2:
                res = (res + SH_C3[0] * y * (3 * xx - yy) * sh[:, 9] + SH_C3[1] * xy * z * sh[:, 10]
                       + SH_C3[2] * y * (4 * zz - xx - yy) * sh[:, 11]
                       + SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12]
                       + SH_C3[4] * x * (4 * zz - xx - yy) * sh[:, 13]
                       + SH_C3[5] * z * (xx - yy) * sh[:, 14] + SH_C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    res = res + 0.5
    return torch.clamp(res, min=0.0)


def deform_project(g: dict, R_f, c_f, s_f, cam: dict, sh_degree: int = 3):
    """Bound-Gaussian deformation + EWA projection.

    g: xyz (N,3), log_scale (N,3), rot (N,4), opacity (N,), sh (N,16,3), binding (N,) long.
    Returns dict with mean2d, depth, conic, opac, rgb, radius(int), rect (N,4 int: x0,y0,x1,y1), visible."""
    b = g["binding"].long()
    Rf, cf, sf = R_f[b], c_f[b], s_f[b]
    mu = torch.einsum("nrc,nc->nr", Rf, g["xyz"]) * sf[:, None] + cf
    Rw = torch.bmm(Rf, quat_to_rotmat(g["rot"]))
    s = torch.exp(g["log_scale"]) * sf[:, None]
    M = Rw * s[:, None, :]
    Sigma = torch.bmm(M, M.transpose(1, 2))
    W = torch.as_tensor(cam["world_to_view"], dtype=mu.dtype)
    t = mu @ W[:3, :3].T + W[:3, 3]
    tz = t[:, 2]
    width, height = cam["width"], cam["height"]
    fx, fy = cam["fl_x"], cam["fl_y"]
    tfx, tfy = cam["tanfovx"], cam["tanfovy"]
    in_front = tz > 0.2
    tzs = torch.where(in_front, tz, torch.ones_like(tz))
    px = fx * t[:, 0] / tzs + (width - 1) * 0.5
    py = fy * t[:, 1] / tzs + (height - 1) * 0.5
    txc = torch.clamp(t[:, 0] / tzs, -1.3 * tfx, 1.3 * tfx) * tzs
    tyc = torch.clamp(t[:, 1] / tzs, -1.3 * tfy, 1.3 * tfy) * tzs
    z = torch.zeros_like(tzs)
    J = torch.stack([fx / tzs, z, -fx * txc / (tzs * tzs), z, fy / tzs, -fy * tyc / (tzs * tzs)], -1).reshape(-1, 2, 3)
    T = J @ W[:3, :3]
    cov = T @ Sigma @ T.transpose(1, 2)
    a = cov[:, 0, 0] + 0.3
    bb = cov[:, 0, 1]
    c = cov[:, 1, 1] + 0.3
    det = a * c - bb * bb
    ok = in_front & (det != 0)
    dets = torch.where(ok, det, torch.ones_like(det))
    conic = torch.stack([c / dets, -bb / dets, a / dets], -1)
    mid = 0.5 * (a + c)
    lam = mid + torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
    radius = torch.ceil(3.0 * torch.sqrt(lam)).detach()
    gx, gy = (width + TILE - 1) // TILE, (height + TILE - 1) // TILE
    pxd, pyd = px.detach(), py.detach()
    x0 = torch.clamp(((pxd - radius) / TILE).to(torch.int32), 0, gx)
    y0 = torch.clamp(((pyd - radius) / TILE).to(torch.int32), 0, gy)
    x1 = torch.clamp(((pxd + radius + (TILE - 1)) / TILE).to(torch.int32), 0, gx)
    y1 = torch.clamp(((pyd + radius + (TILE - 1)) / TILE).to(torch.int32), 0, gy)
    visible = ok & ((x1 - x0) * (y1 - y0) > 0)
    cam_pos = torch.as_tensor(cam["cam_pos"], dtype=mu.dtype)
    dirs = safe_normalize(mu - cam_pos)
    rgb = eval_sh(g["sh"], dirs, sh_degree)
    return {
        "mean2d": torch.stack([px, py], -1), "depth": tz, "conic": conic,
        "opac": torch.sigmoid(g["opacity"]), "rgb": rgb,
        "radius": torch.where(visible, radius, torch.zeros_like(radius)).to(torch.int32),
        "rect": torch.stack([x0, y0, x1, y1], -1), "visible": visible, "mean3d": mu,
    }


# --------------------------------------------------------------------------- binning + composite
def tile_touched(mx, my, A, B, C, o, tx, ty):
    """Engine's tile-inclusion rule (DESIGN.md "Binning"): exact minimum of the conic form over the
    tile's pixel-centre box, with slack, against alpha >= 1/255.  Scalars (python floats)."""
    if not (A > 0 and C > 0):
        return True
    f = torch.tensor
    x0, y0 = float(tx * TILE), float(ty * TILE)
    dxl, dxh, dyl, dyh = mx - (x0 + 15.0), mx - x0, my - (y0 + 15.0), my - y0
    if dxl <= 0 <= dxh and dyl <= 0 <= dyh:
        return True
    nBoC, nBoA = -B / C, -B / A
    clamp = lambda v, lo, hi: min(max(v, lo), hi)
    cands = [(dxl, clamp(nBoC * dxl, dyl, dyh)), (dxh, clamp(nBoC * dxh, dyl, dyh)),
             (clamp(nBoA * dyl, dxl, dxh), dyl), (clamp(nBoA * dyh, dxl, dxh), dyh)]
    best, mag = 3.0e38, 0.0
    for dx, dy in cands:
        t0, t1, t2 = A * dx * dx, 2 * B * dx * dy, C * dy * dy
        q = t0 + t1 + t2
        if q < best:
            best, mag = q, t0 + abs(t1) + t2
    qa = max(best - 4e-5 * mag - 1e-3, 0.0)
    return o * math.exp(-0.5 * qa) >= (1.0 / 255.0) * 0.999


def tile_lists(proj: dict, width: int, height: int, cull: bool = True):
    """Per-tile sorted Gaussian id lists.  Order = ascending (depth bits, id): identical to the
    upstream global (tile<<32|depth) stable radix sort with emission order = Gaussian index.
    cull=True keeps only the tiles of the 3-sigma rectangle that pass tile_touched()."""
    gx, gy = (width + TILE - 1) // TILE, (height + TILE - 1) // TILE
    vis = torch.nonzero(proj["visible"]).squeeze(1)
    rect = proj["rect"][vis]
    depth = proj["depth"].detach()[vis]
    lists = [[] for _ in range(gx * gy)]
    order = torch.argsort(depth.contiguous().view(torch.int32).to(torch.int64) * (1 << 32) + vis, stable=True)
    m2 = proj["mean2d"].detach()[vis].tolist()
    con = proj["conic"].detach()[vis].tolist()
    op = proj["opac"].detach()[vis].tolist()
    for k in order.tolist():
        x0, y0, x1, y1 = rect[k].tolist()
        gid = int(vis[k])
        for ty in range(y0, y1):
            for tx in range(x0, x1):
                if not cull or tile_touched(m2[k][0], m2[k][1], con[k][0], con[k][1], con[k][2], op[k], tx, ty):
                    lists[ty * gx + tx].append(gid)
    return lists


NEAR_TOL = 2e-5     # |255 alpha - 1| below which a (Gaussian, pixel) pair counts as "on the 1/255 threshold"


def composite(proj: dict, lists, width: int, height: int, bg, tiles=None, decide=None, diag=None):
    """Front-to-back alpha compositing, exact order, vectorised per tile.
    Returns image (3,H,W), final_T (H,W), n_contrib (H,W) int32.
    tiles (optional set of tile indices): only these tiles are composited, every other tile shows the background
    (full-size checks of a fixed tile subset: the autograd graph then holds those tiles only).
    decide (optional dict of float32 arrays mean2d (N,2), conic (N,3), opac (N,), e.g. the C oracle's projection): the
    DISCRETE decisions of the algorithm -- which (Gaussian, pixel) pairs reach alpha >= 1/255, where a pixel stops -- are
    taken from these values instead of this module's own projection, whose fp32 geometry differs from the bit-level spec
    by up to ~1e-3 pixel on thin triangles, enough to flip pairs that sit on a threshold.  The composited VALUES (and so
    the autograd graph) still come from `proj`.  The image is a discontinuous function of the parameters at those
    thresholds; a comparison of gradients is only meaningful on one side of them.
    diag (optional dict): receives `near_gaussians` (set of ids) and `near_pixels` (H,W bool): pairs whose alpha lies
    within NEAR_TOL (relative) of 1/255, where two correct fp32 evaluations may still decide differently."""
    gx, gy = (width + TILE - 1) // TILE, (height + TILE - 1) // TILE
    bg = torch.as_tensor(bg, dtype=torch.float32)
    img = torch.zeros(3, gy * TILE, gx * TILE)
    fT = torch.ones(gy * TILE, gx * TILE)
    ncon = torch.zeros(gy * TILE, gx * TILE, dtype=torch.int32)
    yy, xx = torch.meshgrid(torch.arange(TILE, dtype=torch.float32), torch.arange(TILE, dtype=torch.float32), indexing="ij")
    rows = []
    if decide is not None:
        d_m, d_con, d_op = (torch.as_tensor(decide[k], dtype=torch.float32) for k in ("mean2d", "conic", "opac"))
    near_g = set()
    near_px = torch.zeros(gy * TILE, gx * TILE, dtype=torch.bool)
    for ty in range(gy):
        cols = []
        for tx in range(gx):
            ids = lists[ty * gx + tx]
            if tiles is not None and (ty * gx + tx) not in tiles:
                ids = []
            if len(ids) == 0:
                cols.append((bg[:, None, None].expand(3, TILE, TILE), torch.ones(TILE, TILE), torch.zeros(TILE, TILE, dtype=torch.int32)))
                continue
            idt = torch.as_tensor(ids, dtype=torch.long)
            m = proj["mean2d"][idt]
            con = proj["conic"][idt]
            op = proj["opac"][idt]
            col = proj["rgb"][idt]
            px = (xx + tx * TILE).reshape(1, -1)
            py = (yy + ty * TILE).reshape(1, -1)
            dx = m[:, 0:1] - px
            dy = m[:, 1:2] - py
            power = -0.5 * (con[:, 0:1] * dx * dx + con[:, 2:3] * dy * dy) - con[:, 1:2] * dx * dy
            a_raw = op[:, None] * torch.exp(power)
            # value clamped at 0.99, gradient passed straight through (as the upstream rasteriser does)
            alpha = a_raw + (torch.clamp(a_raw, max=0.99) - a_raw).detach()
            if decide is None:
                power_d, alpha_d = power.detach(), alpha.detach()
            else:
                px32, py32 = (xx + tx * TILE).reshape(1, -1), (yy + ty * TILE).reshape(1, -1)
                ddx, ddy = d_m[idt, 0:1] - px32, d_m[idt, 1:2] - py32
                dc = d_con[idt]
                power_d = -0.5 * (dc[:, 0:1] * ddx * ddx + dc[:, 2:3] * ddy * ddy) - dc[:, 1:2] * ddx * ddy
                alpha_d = torch.clamp(d_op[idt][:, None] * torch.exp(power_d), max=0.99)
            keep = (power_d <= 0) & (alpha_d >= 1.0 / 255.0)
            a = torch.where(keep, alpha, torch.zeros_like(alpha))
            Tinc = torch.cumprod(1 - a, dim=0)
            Texc = torch.cat([torch.ones_like(Tinc[:1]), Tinc[:-1]], 0)
            if decide is None:
                stop = (Tinc < 1e-4) & keep
            else:
                stop = (torch.cumprod(1 - torch.where(keep, alpha_d, torch.zeros_like(alpha_d)), dim=0) < 1e-4) & keep
            done = torch.cumsum(stop.to(torch.int32), 0) > 0        # this splat and all later excluded
            if diag is not None:
                near = ((alpha_d * 255.0 - 1.0).abs() < NEAR_TOL) & (power_d <= 0) & ~done
                if bool(near.any()):
                    near_g.update(idt[near.any(1)].tolist())
                    near_px[ty * TILE:(ty + 1) * TILE, tx * TILE:(tx + 1) * TILE] |= near.any(0).reshape(TILE, TILE)
            w = torch.where(done, torch.zeros_like(a), a * Texc)
            C = torch.einsum("np,nc->cp", w, col)
            alive = ~done
            Tfin = torch.where(alive, Tinc, torch.zeros_like(Tinc))
            # final T = T after last non-excluded splat
            n_alive = alive.to(torch.int64).sum(0)
            Tlast = torch.where(n_alive > 0, Tinc.gather(0, (n_alive - 1).clamp(min=0).unsqueeze(0)).squeeze(0), torch.ones_like(Tinc[0]))
            contrib = alive & keep
            idx = torch.arange(1, len(ids) + 1, dtype=torch.int32).unsqueeze(1)
            last = torch.where(contrib, idx, torch.zeros_like(idx)).max(0).values
            out = C + Tlast.unsqueeze(0) * bg[:, None]
            cols.append((out.reshape(3, TILE, TILE), Tlast.reshape(TILE, TILE), last.reshape(TILE, TILE)))
            del Tfin
        rows.append((torch.cat([c[0] for c in cols], 2), torch.cat([c[1] for c in cols], 1), torch.cat([c[2] for c in cols], 1)))
    img = torch.cat([r[0] for r in rows], 1)[:, :height, :width]
    fT = torch.cat([r[1] for r in rows], 0)[:height, :width]
    ncon = torch.cat([r[2] for r in rows], 0)[:height, :width]
    if diag is not None:
        diag["near_gaussians"], diag["near_pixels"] = near_g, near_px[:height, :width]
    return img, fT, ncon


def lists_from_offsets(tile_start, ids):
    """The C oracle's (= the engine's, bit for bit) tile-segmented id array as the per-tile lists composite() takes."""
    ts = [int(x) for x in tile_start]
    return [ids[ts[t]:ts[t + 1]].astype("int64") for t in range(len(ts) - 1)]


def render(rig: dict, g: dict, frame: dict, cam: dict, bg=(0.0, 0.0, 0.0), sh_degree: int = 3, lists=None, tiles=None,
           decide=None, diag=None):
    """One frame: FLAME -> face frames -> deform/project -> bin/sort -> composite.
    lists: per-tile id lists to composite instead of this module's own binning.  The order inside a tile is defined on
    the DEPTH BITS (DESIGN.md "Binning"); this module's fp32 depths differ from the bit-level spec (oracle/splat_oracle.c)
    in the last bit here and there, so two near-equal depths can swap -- handing over the C oracle's lists makes both
    oracles composite in ONE order (the backward comparison then has no order noise).  tiles, decide, diag: see composite()."""
    verts = flame_lbs(rig, frame["shape"], frame["expr"][None], frame["rotmats"][None], frame["translation"][None],
                      frame.get("static_offset"), frame.get("dynamic_offset"))[0]
    R_f, c_f, s_f = face_frames(verts, rig["faces"].long())
    proj = deform_project(g, R_f, c_f, s_f, cam, sh_degree)
    if lists is None:
        lists = tile_lists(proj, cam["width"], cam["height"])
    img, fT, ncon = composite(proj, lists, cam["width"], cam["height"], bg, tiles, decide, diag)
    return {"image": img, "final_T": fT, "n_contrib": ncon, "proj": proj, "lists": lists, "verts": verts,
            "frames": (R_f, c_f, s_f)}


# --------------------------------------------------------------------------- loss
def _gauss_window(size=11, sigma=1.5):
    g = torch.tensor([math.exp(-((x - size // 2) ** 2) / (2.0 * sigma ** 2)) for x in range(size)], dtype=torch.float32)
    return g / g.sum()


def ssim(img1, img2):
    """3DGS SSIM: 11x11 gaussian (sigma 1.5), zero padding, per channel, C1=0.01^2, C2=0.03^2, mean."""
    w1 = _gauss_window()
    w2 = (w1[:, None] * w1[None, :])[None, None].expand(3, 1, 11, 11).contiguous()
    F = torch.nn.functional
    a, b = img1[None], img2[None]
    mu1 = F.conv2d(a, w2, padding=5, groups=3)
    mu2 = F.conv2d(b, w2, padding=5, groups=3)
    s11 = F.conv2d(a * a, w2, padding=5, groups=3) - mu1 * mu1
    s22 = F.conv2d(b * b, w2, padding=5, groups=3) - mu2 * mu2
    s12 = F.conv2d(a * b, w2, padding=5, groups=3) - mu1 * mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    m = ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s11 + s22 + C2))
    return m.mean()


def photometric_loss(img, gt, lambda_dssim=0.2):
    l1 = (img - gt).abs().mean()
    return (1.0 - lambda_dssim) * l1 + lambda_dssim * (1.0 - ssim(img, gt))


def regularisers(g: dict, visible, lambda_xyz=0.01, thr_xyz=1.0, lambda_scale=1.0, thr_scale=0.6):
    """GaussianAvatars local-position and scale regularisers on visible Gaussians
    (SURVEY Appendix A item 8): mean over visible of relu(|xyz|-thr) norm / relu(exp(ls)-thr) norm."""
    vis = visible
    if vis.sum() == 0:
        return torch.zeros(())
    lx = torch.relu(g["xyz"][vis].norm(dim=1) - thr_xyz).mean() * lambda_xyz
    ls = torch.relu(torch.exp(g["log_scale"][vis]) - thr_scale).norm(dim=1).mean() * lambda_scale
    return lx + ls


def adam_step(p, grad, m, v, step, lr, b1=0.9, b2=0.999, eps=1e-15):
    """torch.optim.Adam semantics (no weight decay, no amsgrad), in place on fp32 tensors."""
    m.mul_(b1).add_(grad, alpha=1 - b1)
    v.mul_(b2).addcmul_(grad, grad, value=1 - b2)
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)

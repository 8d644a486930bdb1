/*
 * ORACLE -- test infrastructure, never shipped, never on the product path.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * Plain-C fp32 restatement of the forward hot path with a FIXED operation order (DESIGN.md
 * "Frozen arithmetic"): every step is an individually rounded IEEE-754 binary32 operation or an
 * explicit fmaf, so the HIP kernels can be required to reproduce vertices, triangle frames,
 * radii, tile rectangles and per-tile orders BIT FOR BIT.  Build: gcc -O2 -ffp-contract=off.
 *
 * PARITY UNPINNED: the algorithm (FLAME LBS, triangle-bound Gaussians, EWA splatting) lives in
 * an un-vendored, un-pinned third-party checkout that the reference only launches
 * (02_Visual_Engine/train_ghost.py:27-28,227-271; render_surgery.py:36-37,289-315;
 * .gitignore:27); no golden vector for it exists in the reference.  What is restated is the
 * published algorithm (Kerbl et al. 2023; Qian et al. 2024) under the conventions of
 * SURVEY.md Appendix A, frozen in DESIGN.md.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define TILE 16

static inline float dot3(float ax, float ay, float az, float bx, float by, float bz) {
  return fmaf(az, bz, fmaf(ay, by, ax * bx));
}
static inline float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* Cephes-style expf with fixed operation order (scale activation). */
float orc_exp(float x) {
  x = fminf(fmaxf(x, -87.0f), 88.0f);
  float n = rintf(x * 1.44269504088896341f);
  float r = fmaf(n, -0.693359375f, x);
  r = fmaf(n, 2.12194440e-4f, r);
  float p = 1.9875691500e-4f;
  p = fmaf(p, r, 1.3981999507e-3f);
  p = fmaf(p, r, 8.3334519073e-3f);
  p = fmaf(p, r, 4.1665795894e-2f);
  p = fmaf(p, r, 1.6666665459e-1f);
  p = fmaf(p, r, 5.0000001201e-1f);
  float y = fmaf(p, r * r, r) + 1.0f;
  int e = (int)n;
  return y * u2f((uint32_t)(e + 127) << 23);
}

/* ---- FLAME.  Inputs are exactly the arrays the device reads.
 * v_static [3][v_pad], basis [K][3][V], weights [v_pad][8], j_static [5][3], j_expr [15][n_expr],
 * expr [n_expr], rotmats [5][9], translation [3], dynamic (nullable) [V][3] -> verts [V][3],
 * joint_xf [5][12]. */
void orc_flame_frame(int V, int v_pad, int n_expr, const float* v_static, const float* basis,
                     const float* weights, const float* j_static, const float* j_expr,
                     const float* expr, const float* R, const float* translation,
                     const float* dynamic, float* verts, float* joint_xf) {
  float J[5][3];
  for (int jc = 0; jc < 15; ++jc) {
    float acc = j_static[jc];
    for (int k = 0; k < n_expr; ++k) acc = fmaf(j_expr[jc * n_expr + k], expr[k], acc);
    J[jc / 3][jc % 3] = acc;
  }
  float Rw[5][9], tw[5][3];
  memcpy(Rw[0], R, 9 * sizeof(float));
  for (int c = 0; c < 3; ++c) tw[0][c] = J[0][c];
  for (int j = 1; j < 5; ++j) {
    int p = (j == 1) ? 0 : 1;
    const float* Rl = R + j * 9;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c)
        Rw[j][r * 3 + c] = dot3(Rw[p][r * 3], Rw[p][r * 3 + 1], Rw[p][r * 3 + 2], Rl[c], Rl[3 + c], Rl[6 + c]);
    float dx = J[j][0] - J[p][0], dy = J[j][1] - J[p][1], dz = J[j][2] - J[p][2];
    for (int r = 0; r < 3; ++r) tw[j][r] = dot3(Rw[p][r * 3], Rw[p][r * 3 + 1], Rw[p][r * 3 + 2], dx, dy, dz) + tw[p][r];
  }
  float X[60];
  for (int j = 0; j < 5; ++j) {
    for (int i = 0; i < 9; ++i) X[j * 12 + i] = Rw[j][i];
    for (int r = 0; r < 3; ++r)
      X[j * 12 + 9 + r] = tw[j][r] - dot3(Rw[j][r * 3], Rw[j][r * 3 + 1], Rw[j][r * 3 + 2], J[j][0], J[j][1], J[j][2]);
  }
  if (joint_xf) memcpy(joint_xf, X, sizeof(X));
  const int K = n_expr + 36;
  float* coef = (float*)malloc(sizeof(float) * K);
  for (int k = 0; k < n_expr; ++k) coef[k] = expr[k];
  for (int j = 1; j < 5; ++j)
    for (int i = 0; i < 9; ++i) coef[n_expr + (j - 1) * 9 + i] = R[j * 9 + i] - ((i == 0 || i == 4 || i == 8) ? 1.f : 0.f);
  for (int v = 0; v < V; ++v) {
    float p[3];
    for (int c = 0; c < 3; ++c) {
      float acc = v_static[c * v_pad + v];
      for (int k = 0; k < K; ++k) acc = fmaf(basis[((size_t)k * 3 + c) * V + v], coef[k], acc);
      p[c] = acc;
    }
    const float* w = weights + (size_t)v * 8;
    float M[12];
    for (int i = 0; i < 12; ++i) {
      float m = w[0] * X[i];
      for (int j = 1; j < 5; ++j) m = fmaf(w[j], X[j * 12 + i], m);
      M[i] = m;
    }
    float ox = dot3(M[0], M[1], M[2], p[0], p[1], p[2]) + M[9];
    float oy = dot3(M[3], M[4], M[5], p[0], p[1], p[2]) + M[10];
    float oz = dot3(M[6], M[7], M[8], p[0], p[1], p[2]) + M[11];
    if (dynamic) { ox += dynamic[v * 3]; oy += dynamic[v * 3 + 1]; oz += dynamic[v * 3 + 2]; }
    verts[v * 3] = ox + translation[0];
    verts[v * 3 + 1] = oy + translation[1];
    verts[v * 3 + 2] = oz + translation[2];
  }
  free(coef);
}

static inline void safe_normalize3(float* x, float* y, float* z) {
  float d = fmaxf(dot3(*x, *y, *z, *x, *y, *z), 1e-20f);
  float l = sqrtf(d);
  *x = *x / l; *y = *y / l; *z = *z / l;
}

/* Triangle frames (Appendix A item 2): face_xf [F][16] = R row-major (columns a0,n,a2), centre, scale, 0,0,0 */
void orc_face_frames(int F, const float* verts, const int32_t* faces, float* face_xf) {
  for (int f = 0; f < F; ++f) {
    const float* v0 = verts + 3 * (size_t)faces[f * 3];
    const float* v1 = verts + 3 * (size_t)faces[f * 3 + 1];
    const float* v2 = verts + 3 * (size_t)faces[f * 3 + 2];
    float e1x = v1[0] - v0[0], e1y = v1[1] - v0[1], e1z = v1[2] - v0[2];
    float e2x = v2[0] - v0[0], e2y = v2[1] - v0[1], e2z = v2[2] - v0[2];
    float a0x = e1x, a0y = e1y, a0z = e1z;
    safe_normalize3(&a0x, &a0y, &a0z);
    float nx = fmaf(a0y, e2z, -(a0z * e2y)), ny = fmaf(a0z, e2x, -(a0x * e2z)), nz = fmaf(a0x, e2y, -(a0y * e2x));
    safe_normalize3(&nx, &ny, &nz);
    float cx = fmaf(ny, a0z, -(nz * a0y)), cy = fmaf(nz, a0x, -(nx * a0z)), cz = fmaf(nx, a0y, -(ny * a0x));
    safe_normalize3(&cx, &cy, &cz);
    float a2x = -cx, a2y = -cy, a2z = -cz;
    float s0 = sqrtf(dot3(e1x, e1y, e1z, e1x, e1y, e1z));
    float s1 = fabsf(dot3(a2x, a2y, a2z, e2x, e2y, e2z));
    const float third = 1.0f / 3.0f;
    float* o = face_xf + (size_t)f * 16;
    o[0] = a0x; o[1] = nx; o[2] = a2x; o[3] = a0y; o[4] = ny; o[5] = a2y; o[6] = a0z; o[7] = nz; o[8] = a2z;
    o[9] = ((v0[0] + v1[0]) + v2[0]) * third; o[10] = ((v0[1] + v1[1]) + v2[1]) * third; o[11] = ((v0[2] + v1[2]) + v2[2]) * third;
    o[12] = (s0 + s1) * 0.5f; o[13] = o[14] = o[15] = 0.f;
  }
}

typedef struct {
  float view[12];
  float cam_pos[3];
  float fx, fy, cx, cy, limx, limy;
  int width, height, sh_degree;
  float bg[3];
} orc_camera;

static const float SH_C0 = 0.28209479177387814f, SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                               -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};

/* Deform + project (Appendix A items 3-4).  params [59][n_pad]; outputs per Gaussian:
 * mean2d[2], conic[3], opac, rgb[3], depth, radius (int, 0 = culled), rect[4], clamp bits. */
void orc_project(int n, int n_pad, const float* params, const int32_t* binding, const float* face_xf,
                 const orc_camera* cam, float* mean2d, float* conic, float* opac, float* rgb, float* depth,
                 int32_t* radius, int32_t* rect, int32_t* clampbits) {
  const int gx = (cam->width + TILE - 1) / TILE, gy = (cam->height + TILE - 1) / TILE;
  const float* W = cam->view;
  for (int i = 0; i < n; ++i) {
#define P(pl) params[(size_t)(pl)*n_pad + i]
    const float* fr = face_xf + (size_t)binding[i] * 16;
    const float R00 = fr[0], R01 = fr[1], R02 = fr[2], R10 = fr[3], R11 = fr[4], R12 = fr[5], R20 = fr[6], R21 = fr[7], R22 = fr[8];
    const float cfx = fr[9], cfy = fr[10], cfz = fr[11], sf = fr[12];
    const float lx = P(0), ly = P(1), lz = P(2);
    const float mx = fmaf(dot3(R00, R01, R02, lx, ly, lz), sf, cfx);
    const float my = fmaf(dot3(R10, R11, R12, lx, ly, lz), sf, cfy);
    const float mz = fmaf(dot3(R20, R21, R22, lx, ly, lz), sf, cfz);
    const float tx = dot3(W[0], W[1], W[2], mx, my, mz) + W[3];
    const float ty = dot3(W[4], W[5], W[6], mx, my, mz) + W[7];
    const float tz = dot3(W[8], W[9], W[10], mx, my, mz) + W[11];
    mean2d[i * 2] = mean2d[i * 2 + 1] = 0.f; conic[i * 3] = conic[i * 3 + 1] = conic[i * 3 + 2] = 0.f;
    opac[i] = 0.f; rgb[i * 3] = rgb[i * 3 + 1] = rgb[i * 3 + 2] = 0.f; depth[i] = 0.f; radius[i] = 0;
    rect[i * 4] = rect[i * 4 + 1] = rect[i * 4 + 2] = rect[i * 4 + 3] = 0; clampbits[i] = 0;
    if (!(tz > 0.2f)) continue;
    float qw = P(6), qx = P(7), qy = P(8), qz = P(9);
    const float qn = sqrtf(fmaf(qz, qz, fmaf(qy, qy, fmaf(qx, qx, qw * qw))));
    qw = qw / qn; qx = qx / qn; qy = qy / qn; qz = qz / qn;
    const float Q00 = 1.f - 2.f * fmaf(qy, qy, qz * qz), Q01 = 2.f * fmaf(qx, qy, -(qw * qz)), Q02 = 2.f * fmaf(qx, qz, qw * qy);
    const float Q10 = 2.f * fmaf(qx, qy, qw * qz), Q11 = 1.f - 2.f * fmaf(qx, qx, qz * qz), Q12 = 2.f * fmaf(qy, qz, -(qw * qx));
    const float Q20 = 2.f * fmaf(qx, qz, -(qw * qy)), Q21 = 2.f * fmaf(qy, qz, qw * qx), Q22 = 1.f - 2.f * fmaf(qx, qx, qy * qy);
    const float s0 = orc_exp(P(3)) * sf, s1 = orc_exp(P(4)) * sf, s2 = orc_exp(P(5)) * sf;
    const float M00 = dot3(R00, R01, R02, Q00, Q10, Q20) * s0, M01 = dot3(R00, R01, R02, Q01, Q11, Q21) * s1, M02 = dot3(R00, R01, R02, Q02, Q12, Q22) * s2;
    const float M10 = dot3(R10, R11, R12, Q00, Q10, Q20) * s0, M11 = dot3(R10, R11, R12, Q01, Q11, Q21) * s1, M12 = dot3(R10, R11, R12, Q02, Q12, Q22) * s2;
    const float M20 = dot3(R20, R21, R22, Q00, Q10, Q20) * s0, M21 = dot3(R20, R21, R22, Q01, Q11, Q21) * s1, M22 = dot3(R20, R21, R22, Q02, Q12, Q22) * s2;
    const float S00 = dot3(M00, M01, M02, M00, M01, M02), S01 = dot3(M00, M01, M02, M10, M11, M12), S02 = dot3(M00, M01, M02, M20, M21, M22);
    const float S11 = dot3(M10, M11, M12, M10, M11, M12), S12 = dot3(M10, M11, M12, M20, M21, M22), S22 = dot3(M20, M21, M22, M20, M21, M22);
    const float xz = tx / tz, yz = ty / tz;
    const float px = fmaf(cam->fx, xz, cam->cx), py = fmaf(cam->fy, yz, cam->cy);
    const float txc = clampf(xz, -cam->limx, cam->limx) * tz, tyc = clampf(yz, -cam->limy, cam->limy) * tz;
    const float tz2 = tz * tz;
    const float J00 = cam->fx / tz, J02 = -(cam->fx * txc) / tz2, J11 = cam->fy / tz, J12 = -(cam->fy * tyc) / tz2;
    const float T00 = fmaf(J02, W[8], J00 * W[0]), T01 = fmaf(J02, W[9], J00 * W[1]), T02 = fmaf(J02, W[10], J00 * W[2]);
    const float T10 = fmaf(J12, W[8], J11 * W[4]), T11 = fmaf(J12, W[9], J11 * W[5]), T12 = fmaf(J12, W[10], J11 * W[6]);
    const float u0 = dot3(S00, S01, S02, T00, T01, T02), u1 = dot3(S01, S11, S12, T00, T01, T02), u2 = dot3(S02, S12, S22, T00, T01, T02);
    const float w0 = dot3(S00, S01, S02, T10, T11, T12), w1 = dot3(S01, S11, S12, T10, T11, T12), w2 = dot3(S02, S12, S22, T10, T11, T12);
    const float a = dot3(T00, T01, T02, u0, u1, u2) + 0.3f;
    const float b = dot3(T10, T11, T12, u0, u1, u2);
    const float c = dot3(T10, T11, T12, w0, w1, w2) + 0.3f;
    const float det = fmaf(a, c, -(b * b));
    if (det == 0.f) continue;
    const float mid = 0.5f * (a + c);
    const float lam = mid + sqrtf(fmaxf(0.1f, fmaf(mid, mid, -det)));
    const float rad = fminf(ceilf(3.f * sqrtf(lam)), 1048575.f);
    int x0 = (int)clampf((px - rad) / 16.f, -1.f, 4096.f), y0 = (int)clampf((py - rad) / 16.f, -1.f, 4096.f);
    int x1 = (int)clampf(((px + rad) + 15.f) / 16.f, -1.f, 4096.f), y1 = (int)clampf(((py + rad) + 15.f) / 16.f, -1.f, 4096.f);
    x0 = x0 < 0 ? 0 : (x0 > gx ? gx : x0); x1 = x1 < 0 ? 0 : (x1 > gx ? gx : x1);
    y0 = y0 < 0 ? 0 : (y0 > gy ? gy : y0); y1 = y1 < 0 ? 0 : (y1 > gy ? gy : y1);
    if ((x1 - x0) * (y1 - y0) <= 0) continue;
    /* colour: tolerance-level from here */
    float dx = mx - cam->cam_pos[0], dy = my - cam->cam_pos[1], dz = mz - cam->cam_pos[2];
    const float dl = sqrtf(fmaxf(dot3(dx, dy, dz, dx, dy, dz), 1e-20f));
    dx /= dl; dy /= dl; dz /= dl;
    int cb = 0;
    for (int ch = 0; ch < 3; ++ch) {
#define S(k) P(11 + 3 * (k) + ch)
      float r = SH_C0 * S(0);
      if (cam->sh_degree > 0) {
        r = r - SH_C1 * dy * S(1) + SH_C1 * dz * S(2) - SH_C1 * dx * S(3);
        if (cam->sh_degree > 1) {
          const float xx = dx * dx, yy = dy * dy, zz = dz * dz, xy = dx * dy, yzz = dy * dz, xzz = dx * dz;
          r = r + SH_C2[0] * xy * S(4) + SH_C2[1] * yzz * S(5) + SH_C2[2] * (2.f * zz - xx - yy) * S(6) + SH_C2[3] * xzz * S(7) + SH_C2[4] * (xx - yy) * S(8);
          if (cam->sh_degree > 2)
            r = r + SH_C3[0] * dy * (3.f * xx - yy) * S(9) + SH_C3[1] * xy * dz * S(10) + SH_C3[2] * dy * (4.f * zz - xx - yy) * S(11) +
                SH_C3[3] * dz * (2.f * zz - 3.f * xx - 3.f * yy) * S(12) + SH_C3[4] * dx * (4.f * zz - xx - yy) * S(13) +
                SH_C3[5] * dz * (xx - yy) * S(14) + SH_C3[6] * dx * (xx - 3.f * yy) * S(15);
        }
      }
      r += 0.5f;
      if (r < 0.f) { cb |= 1 << ch; r = 0.f; }
      rgb[i * 3 + ch] = r;
#undef S
    }
    mean2d[i * 2] = px; mean2d[i * 2 + 1] = py;
    conic[i * 3] = c / det; conic[i * 3 + 1] = -b / det; conic[i * 3 + 2] = a / det;
    opac[i] = 1.f / (1.f + orc_exp(-P(10)));   /* frozen: feeds the tile test */
    depth[i] = tz; radius[i] = (int32_t)rad;
    rect[i * 4] = x0; rect[i * 4 + 1] = y0; rect[i * 4 + 2] = x1; rect[i * 4 + 3] = y1;
    clampbits[i] = cb;
#undef P
  }
}

/* Frozen tile-inclusion test (DESIGN.md "Binning"): can the splat reach alpha >= 1/255 at a pixel
 * centre of tile (tx,ty)?  Same operations in the same order as csrc/common.hpp tile_touched(). */
int orc_tile_touched(float mx, float my, float A, float B, float C, float o, int tx, int ty) {
  if (!(A > 0.f && C > 0.f)) return 1;
  const float x0 = (float)(tx * TILE), y0 = (float)(ty * TILE);
  const float dxl = mx - (x0 + 15.f), dxh = mx - x0, dyl = my - (y0 + 15.f), dyh = my - y0;
  if (dxl <= 0.f && dxh >= 0.f && dyl <= 0.f && dyh >= 0.f) return 1;
  const float nBoC = -B / C, nBoA = -B / A;
  float best = 3.0e38f, mag = 0.f;
  const float cx[4] = {dxl, dxh, fminf(fmaxf(nBoA * dyl, dxl), dxh), fminf(fmaxf(nBoA * dyh, dxl), dxh)};
  const float cy[4] = {fminf(fmaxf(nBoC * dxl, dyl), dyh), fminf(fmaxf(nBoC * dxh, dyl), dyh), dyl, dyh};
  for (int k = 0; k < 4; ++k) {
    const float t0 = (A * cx[k]) * cx[k], t1 = ((2.f * B) * cx[k]) * cy[k], t2 = (C * cy[k]) * cy[k];
    const float q = (t0 + t1) + t2;
    if (q < best) { best = q; mag = (t0 + fabsf(t1)) + t2; }
  }
  const float qa = fmaxf((best - 4e-5f * mag) - 1e-3f, 0.f);
  return o * orc_exp(-0.5f * qa) >= (1.f / 255.f) * 0.999f;
}

typedef struct { uint32_t depth, id; } orc_pair;
static int pair_cmp(const void* a, const void* b) {
  const orc_pair* x = (const orc_pair*)a; const orc_pair* y = (const orc_pair*)b;
  if (x->depth != y->depth) return x->depth < y->depth ? -1 : 1;
  return x->id < y->id ? -1 : (x->id > y->id ? 1 : 0);
}

/* Binning + per-tile order (Appendix A item 5).  A (Gaussian, tile) pair exists for every tile of
 * the 3-sigma rectangle that passes orc_tile_touched() (cull != 0) -- or for every tile of the
 * rectangle (cull == 0, the upstream rule; used by tests to show the image does not depend on it).
 * tile_start [n_tiles+1], sorted_ids [D] (caller sizes it from the returned D of a first call
 * with sorted_ids == NULL). */
int64_t orc_bin_sort(int n, int width, int height, const float* depth, const int32_t* radius, const int32_t* rect,
                     const float* mean2d, const float* conic, const float* opac, int cull,
                     uint32_t* tile_start, uint32_t* sorted_ids) {
#define TOUCH(i, x, y) (!cull || orc_tile_touched(mean2d[(i)*2], mean2d[(i)*2 + 1], conic[(i)*3], conic[(i)*3 + 1], conic[(i)*3 + 2], opac[i], x, y))
  const int gx = (width + TILE - 1) / TILE, gy = (height + TILE - 1) / TILE, nt = gx * gy;
  uint32_t* cnt = (uint32_t*)calloc((size_t)nt + 1, 4);
  for (int i = 0; i < n; ++i) {
    if (radius[i] <= 0) continue;
    for (int y = rect[i * 4 + 1]; y < rect[i * 4 + 3]; ++y)
      for (int x = rect[i * 4]; x < rect[i * 4 + 2]; ++x)
        if (TOUCH(i, x, y)) cnt[y * gx + x]++;
  }
  int64_t D = 0;
  for (int t = 0; t < nt; ++t) { tile_start[t] = (uint32_t)D; D += cnt[t]; }
  tile_start[nt] = (uint32_t)D;
  if (!sorted_ids) { free(cnt); return D; }
  orc_pair* pairs = (orc_pair*)malloc(sizeof(orc_pair) * (size_t)(D > 0 ? D : 1));
  memset(cnt, 0, (size_t)nt * 4);
  for (int i = 0; i < n; ++i) {
    if (radius[i] <= 0) continue;
    for (int y = rect[i * 4 + 1]; y < rect[i * 4 + 3]; ++y)
      for (int x = rect[i * 4]; x < rect[i * 4 + 2]; ++x) {
        if (!TOUCH(i, x, y)) continue;
        int t = y * gx + x;
        orc_pair p = {f2u(depth[i]), (uint32_t)i};
        pairs[tile_start[t] + cnt[t]++] = p;
      }
  }
  for (int t = 0; t < nt; ++t) {
    qsort(pairs + tile_start[t], cnt[t], sizeof(orc_pair), pair_cmp);
    for (uint32_t k = 0; k < cnt[t]; ++k) sorted_ids[tile_start[t] + k] = pairs[tile_start[t] + k].id;
  }
  free(pairs); free(cnt);
  return D;
#undef TOUCH
}

/* Front-to-back composite (Appendix A item 6). image [3][H][W], final_T [H][W], n_contrib [H][W].
 * near (optional, [H][W] bytes): which pixels own a DISCRETE decision that two correct fp32 evaluations may take differently --
 *   bit 0: a pair the pixel evaluated before it stopped has |255 alpha - 1| < tol_alpha (the 1/255 inclusion threshold),
 *   bit 1: a pair's T(1 - alpha) lies within tol_T (relative) of the 1e-4 stop threshold.
 * Every other pixel ("calm") must show the same n_contrib in any implementation of the spec and the same colour up to
 * rounding; tests state the bound.  The values the function returns do not depend on `near`. */
static void composite_impl(int width, int height, const float* bg, const uint32_t* tile_start, const uint32_t* sorted_ids,
                           const float* mean2d, const float* conic, const float* opac, const float* rgb, float* image,
                           float* final_T, uint32_t* n_contrib, uint8_t* near, float tol_alpha, float tol_T) {
  const int gx = (width + TILE - 1) / TILE;
  const size_t plane = (size_t)width * height;
#pragma omp parallel for schedule(dynamic, 4)
  for (int py = 0; py < height; ++py)
    for (int px = 0; px < width; ++px) {
      const int t = (py / TILE) * gx + (px / TILE);
      const float fx = (float)px, fy = (float)py;
      float T = 1.f, C0 = 0.f, C1 = 0.f, C2 = 0.f;
      uint32_t contributor = 0, last = 0;
      uint8_t flags = 0;
      for (uint32_t k = tile_start[t]; k < tile_start[t + 1]; ++k) {
        const uint32_t id = sorted_ids[k];
        ++contributor;
        const float dx = mean2d[id * 2] - fx, dy = mean2d[id * 2 + 1] - fy;
        const float A = conic[id * 3], B = conic[id * 3 + 1], Cc = conic[id * 3 + 2];
        const float power = fmaf(-0.5f, fmaf(A * dx, dx, Cc * dy * dy), -(B * dx) * dy);
        if (power > 0.f) continue;
        const float alpha = fminf(0.99f, opac[id] * expf(power));
        if (fabsf(alpha * 255.f - 1.f) < tol_alpha) flags |= 1;
        if (alpha < (1.f / 255.f)) continue;
        const float Tn = T * (1.f - alpha);
        if (fabsf(Tn - 1e-4f) < tol_T * 1e-4f) flags |= 2;
        if (Tn < 1e-4f) break;
        const float w = alpha * T;
        C0 = fmaf(rgb[id * 3], w, C0); C1 = fmaf(rgb[id * 3 + 1], w, C1); C2 = fmaf(rgb[id * 3 + 2], w, C2);
        T = Tn;
        last = contributor;
      }
      const size_t o = (size_t)py * width + px;
      image[o] = fmaf(T, bg[0], C0); image[plane + o] = fmaf(T, bg[1], C1); image[2 * plane + o] = fmaf(T, bg[2], C2);
      final_T[o] = T; n_contrib[o] = last;
      if (near) near[o] = flags;
    }
}

void orc_composite(int width, int height, const float* bg, const uint32_t* tile_start, const uint32_t* sorted_ids,
                   const float* mean2d, const float* conic, const float* opac, const float* rgb, float* image,
                   float* final_T, uint32_t* n_contrib) {
  composite_impl(width, height, bg, tile_start, sorted_ids, mean2d, conic, opac, rgb, image, final_T, n_contrib, NULL, 0.f, 0.f);
}

void orc_composite_diag(int width, int height, const float* bg, const uint32_t* tile_start, const uint32_t* sorted_ids,
                        const float* mean2d, const float* conic, const float* opac, const float* rgb, float* image,
                        float* final_T, uint32_t* n_contrib, uint8_t* near, float tol_alpha, float tol_T) {
  composite_impl(width, height, bg, tile_start, sorted_ids, mean2d, conic, opac, rgb, image, final_T, n_contrib, near, tol_alpha, tol_T);
}

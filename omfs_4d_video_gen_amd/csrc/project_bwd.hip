// Per-Gaussian backward: 2D-splat gradient record -> gradients of the 59 stored parameters.
// One lane per Gaussian; reads the 64-byte dsplat record, the parameters and the face record,
// recomputes the forward intermediates in registers (cheaper than storing them: 59 planes are
// streamed anyway) and writes 59 coalesced planes.  The GaussianAvatars position/scale
// regularisers are folded in here, so no extra pass over the
// parameters exists.  Algorithmic bytes per Gaussian: 64 + 240 + 236 (+ face record, L2).
//
// Spec: SURVEY.md Appendix A item 7 (projection / deformation backward).  With gb->dface the gradient
// w.r.t. the parent triangle's frame record is accumulated as well (FLAME fine-tuning: flame.hip,
// omfs_face_frames_bwd / omfs_flame_skin_bwd).
#include "common.hpp"

namespace omfs {

struct ProjCamB {
  float view[12];
  float cam_pos[3];
  float fx, fy, limx, limy;
  int sh_degree;
};

struct RegK {
  float lambda_xyz, thr_xyz, lambda_scale, thr_scale;
};

// 180 VGPRs as the compiler allocates them leave two waves per SIMD for a kernel that waits on ~110 loads per lane; asking for
// three costs four spilled words and buys latency hiding (0.057 -> 0.049 ms at 300 k Gaussians; four waves spill 50 words: 0.077)
#ifndef OMFS_PBWD_WAVES
#define OMFS_PBWD_WAVES 3
#endif
#define OMFS_PBWD_ATTR __attribute__((amdgpu_waves_per_eu(OMFS_PBWD_WAVES, 8)))
__global__ __launch_bounds__(256) OMFS_PBWD_ATTR void project_bwd_kernel(int n, int n_pad, const float* __restrict__ params,
                                                          const int32_t* __restrict__ binding,
                                                          const float* __restrict__ face_xf, ProjCamB cam,
                                                          const float4* __restrict__ g0, const float4* __restrict__ g1,
                                                          const float4* __restrict__ g2,
                                                          float4* __restrict__ dsplat, RegK reg,
                                                          const uint32_t* __restrict__ n_visible,
                                                          float* __restrict__ grads, float* __restrict__ densify_stats,
                                                          float half_w, float half_h, float* __restrict__ dface, float* __restrict__ drgb_out,
                                                          float* __restrict__ dir_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  auto P = [&](int plane) { return params[(size_t)plane * n_pad + i]; };
  auto G = [&](int plane, float v) { grads[(size_t)plane * n_pad + i] = v; };

  const uint32_t rbits = __float_as_uint(g2[RI(i)].z);
  const bool visible = (rbits & 0xFFFFFu) != 0u;
  if (!visible) {
    for (int p = 0; p < (drgb_out ? OMFS_P_SH + 3 : OMFS_NPLANES); ++p) G(p, 0.f);
    if (drgb_out)
      for (int ch = 0; ch < 3; ++ch) drgb_out[(size_t)ch * n_pad + i] = 0.f;
    if (dir_out)
      for (int ch = 0; ch < 3; ++ch) dir_out[(size_t)ch * n_pad + i] = 0.f;
    if (dface) {
      float4* o = reinterpret_cast<float4*>(dface) + (size_t)i * 4;
      for (int q = 0; q < 4; ++q) o[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    return;
  }
  const uint32_t clampbits = rbits >> 28;
  const float4 d0 = dsplat[(size_t)i * 4 + 0], d1 = dsplat[(size_t)i * 4 + 1], d2 = dsplat[(size_t)i * 4 + 2];
  {   // consumed: leave the record zeroed for the next iteration's composite_bwd (no separate clear pass)
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    dsplat[(size_t)i * 4 + 0] = z; dsplat[(size_t)i * 4 + 1] = z; dsplat[(size_t)i * 4 + 2] = z;
  }
  // composite_bwd accumulates the MOMENTS of dL/dG * G over the pixels (dx = mean - pixel):
  //   d0 = (S_x, S_y, S_xx, S_xy), d1.x = S_yy;   with the projected conic (A, B, C) of g0 / g1
  //   d mean2d = -(A S_x + B S_y, C S_y + B S_x),   d conic = (-S_xx / 2, -S_xy, -S_yy / 2)
  const float4 c0 = g0[RI(i)];
  const float cA = c0.z, cB = c0.w, cC = g1[RI(i)].x;
  const float dpx = -fma_(cA, d0.x, cB * d0.y), dpy = -fma_(cC, d0.y, cB * d0.x);
  const float dA = -0.5f * d0.z, dB = -d0.w, dC = -0.5f * d1.x, dop = d1.y;
  if (densify_stats) {   // adaptive density control statistics
    const float gx = dpx * half_w, gy = dpy * half_h;
    densify_stats[i] += sqrtf(gx * gx + gy * gy);
    densify_stats[(size_t)n_pad + i] += 1.f;
  }
  float drgb[3] = {d1.z, d1.w, d2.x};

  const float4* fr = reinterpret_cast<const float4*>(face_xf) + (size_t)binding[i] * 4;
  const float4 f0 = fr[0], f1 = fr[1], f2 = fr[2], f3 = fr[3];
  const float Rf[9] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w, f2.x};
  const float cf[3] = {f2.y, f2.z, f2.w};
  const float sf = f3.x;

  // ---- forward recompute
  const float l[3] = {P(OMFS_P_XYZ + 0), P(OMFS_P_XYZ + 1), P(OMFS_P_XYZ + 2)};
  float mu[3];
  for (int r = 0; r < 3; ++r) mu[r] = fma_(dot3_(Rf[r * 3], Rf[r * 3 + 1], Rf[r * 3 + 2], l[0], l[1], l[2]), sf, cf[r]);
  const float* W = cam.view;
  const float tx = dot3_(W[0], W[1], W[2], mu[0], mu[1], mu[2]) + W[3];
  const float ty = dot3_(W[4], W[5], W[6], mu[0], mu[1], mu[2]) + W[7];
  const float tz = dot3_(W[8], W[9], W[10], mu[0], mu[1], mu[2]) + W[11];
  float q[4] = {P(OMFS_P_ROT + 0), P(OMFS_P_ROT + 1), P(OMFS_P_ROT + 2), P(OMFS_P_ROT + 3)};
  const float qn = sqrtf(fma_(q[3], q[3], fma_(q[2], q[2], fma_(q[1], q[1], q[0] * q[0]))));
  const float qw = q[0] / qn, qx = q[1] / qn, qy = q[2] / qn, qz = q[3] / qn;
  const float Q[9] = {1.f - 2.f * fma_(qy, qy, qz * qz), 2.f * fma_(qx, qy, -(qw * qz)), 2.f * fma_(qx, qz, qw * qy),
                      2.f * fma_(qx, qy, qw * qz), 1.f - 2.f * fma_(qx, qx, qz * qz), 2.f * fma_(qy, qz, -(qw * qx)),
                      2.f * fma_(qx, qz, -(qw * qy)), 2.f * fma_(qy, qz, qw * qx), 1.f - 2.f * fma_(qx, qx, qy * qy)};
  const float ls[3] = {P(OMFS_P_SCALE + 0), P(OMFS_P_SCALE + 1), P(OMFS_P_SCALE + 2)};
  const float es[3] = {exp_exact(ls[0]), exp_exact(ls[1]), exp_exact(ls[2])};
  const float s[3] = {es[0] * sf, es[1] * sf, es[2] * sf};
  float Rw[9], M[9];
  for (int r = 0; r < 3; ++r)
    for (int k = 0; k < 3; ++k) {
      Rw[r * 3 + k] = dot3_(Rf[r * 3], Rf[r * 3 + 1], Rf[r * 3 + 2], Q[k], Q[3 + k], Q[6 + k]);
      M[r * 3 + k] = Rw[r * 3 + k] * s[k];
    }
  float S[9];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) S[r * 3 + c] = dot3_(M[r * 3], M[r * 3 + 1], M[r * 3 + 2], M[c * 3], M[c * 3 + 1], M[c * 3 + 2]);
  const float xz = tx / tz, yz = ty / tz;
  const bool clx = (xz < -cam.limx) || (xz > cam.limx), cly = (yz < -cam.limy) || (yz > cam.limy);
  const float txc = fminf(fmaxf(xz, -cam.limx), cam.limx) * tz, tyc = fminf(fmaxf(yz, -cam.limy), cam.limy) * tz;
  const float tz2 = tz * tz;
  const float J00 = cam.fx / tz, J02 = -(cam.fx * txc) / tz2, J11 = cam.fy / tz, J12 = -(cam.fy * tyc) / tz2;
  const float T0[3] = {fma_(J02, W[8], J00 * W[0]), fma_(J02, W[9], J00 * W[1]), fma_(J02, W[10], J00 * W[2])};
  const float T1[3] = {fma_(J12, W[8], J11 * W[4]), fma_(J12, W[9], J11 * W[5]), fma_(J12, W[10], J11 * W[6])};
  float u[3], w[3];
  for (int r = 0; r < 3; ++r) {
    u[r] = dot3_(S[r * 3], S[r * 3 + 1], S[r * 3 + 2], T0[0], T0[1], T0[2]);
    w[r] = dot3_(S[r * 3], S[r * 3 + 1], S[r * 3 + 2], T1[0], T1[1], T1[2]);
  }
  const float a = dot3_(T0[0], T0[1], T0[2], u[0], u[1], u[2]) + 0.3f;
  const float b = dot3_(T1[0], T1[1], T1[2], u[0], u[1], u[2]);
  const float c = dot3_(T1[0], T1[1], T1[2], w[0], w[1], w[2]) + 0.3f;
  const float det = fma_(a, c, -(b * b));
  const float id2 = 1.f / (det * det);

  // ---- conic -> cov2d
  const float da = (-c * c * dA + b * c * dB - b * b * dC) * id2;
  const float db = (2.f * b * c * dA - (det + 2.f * b * b) * dB + 2.f * a * b * dC) * id2;
  const float dc = (-b * b * dA + a * b * dB - a * a * dC) * id2;

  // ---- cov2d -> Sigma (symmetrised), T
  float dM[9];
  {
    float Gs[9];
    for (int r = 0; r < 3; ++r)
      for (int k = 0; k < 3; ++k)
        Gs[r * 3 + k] = 2.f * da * T0[r] * T0[k] + db * (T0[r] * T1[k] + T1[r] * T0[k]) + 2.f * dc * T1[r] * T1[k];
    for (int r = 0; r < 3; ++r)
      for (int k = 0; k < 3; ++k) dM[r * 3 + k] = Gs[r * 3] * M[k] + Gs[r * 3 + 1] * M[3 + k] + Gs[r * 3 + 2] * M[6 + k];
  }
  float dT0[3], dT1[3];
  for (int r = 0; r < 3; ++r) {
    dT0[r] = 2.f * da * u[r] + db * w[r];
    dT1[r] = 2.f * dc * w[r] + db * u[r];
  }
  const float dJ00 = dT0[0] * W[0] + dT0[1] * W[1] + dT0[2] * W[2];
  const float dJ02 = dT0[0] * W[8] + dT0[1] * W[9] + dT0[2] * W[10];
  const float dJ11 = dT1[0] * W[4] + dT1[1] * W[5] + dT1[2] * W[6];
  const float dJ12 = dT1[0] * W[8] + dT1[1] * W[9] + dT1[2] * W[10];
  const float itz = 1.f / tz, itz2 = itz * itz;
  float dtx = dpx * cam.fx * itz, dty = dpy * cam.fy * itz;
  float dtz = -(dpx * cam.fx * tx + dpy * cam.fy * ty) * itz2 - (dJ00 * cam.fx + dJ11 * cam.fy) * itz2;
  if (clx) dtz += dJ02 * (-J02 * itz);
  else { dtx += dJ02 * (-cam.fx * itz2); dtz += dJ02 * (2.f * cam.fx * tx * itz2 * itz); }
  if (cly) dtz += dJ12 * (-J12 * itz);
  else { dty += dJ12 * (-cam.fy * itz2); dtz += dJ12 * (2.f * cam.fy * ty * itz2 * itz); }
  float dmu[3] = {W[0] * dtx + W[4] * dty + W[8] * dtz, W[1] * dtx + W[5] * dty + W[9] * dtz,
                  W[2] * dtx + W[6] * dty + W[10] * dtz};

  // ---- colour
  float vx = mu[0] - cam.cam_pos[0], vy = mu[1] - cam.cam_pos[1], vz = mu[2] - cam.cam_pos[2];
  const float vl = sqrtf(fmaxf(vx * vx + vy * vy + vz * vz, 1e-20f));
  const float x = vx / vl, y = vy / vl, z = vz / vl;
  for (int ch = 0; ch < 3; ++ch)
    if ((clampbits >> ch) & 1u) drgb[ch] = 0.f;
  if (drgb_out)
    for (int ch = 0; ch < 3; ++ch) drgb_out[(size_t)ch * n_pad + i] = drgb[ch];
  if (dir_out) { dir_out[i] = x; dir_out[(size_t)n_pad + i] = y; dir_out[(size_t)2 * n_pad + i] = z; }
  {
    constexpr float C0 = 0.28209479177387814f, C1 = 0.4886025119029199f;
    constexpr float C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f, 0.5462742152960396f};
    constexpr float C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                             -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yzp = y * z, xzp = x * z;
    float basis[16];
    basis[0] = C0;
    basis[1] = -C1 * y; basis[2] = C1 * z; basis[3] = -C1 * x;
    basis[4] = C2[0] * xy; basis[5] = C2[1] * yzp; basis[6] = C2[2] * (2.f * zz - xx - yy); basis[7] = C2[3] * xzp; basis[8] = C2[4] * (xx - yy);
    basis[9] = C3[0] * y * (3.f * xx - yy); basis[10] = C3[1] * xy * z; basis[11] = C3[2] * y * (4.f * zz - xx - yy);
    basis[12] = C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy); basis[13] = C3[4] * x * (4.f * zz - xx - yy);
    basis[14] = C3[5] * z * (xx - yy); basis[15] = C3[6] * x * (xx - 3.f * yy);
    const int ncoef = (cam.sh_degree + 1) * (cam.sh_degree + 1);
    float ddx = 0.f, ddy = 0.f, ddz = 0.f;
    for (int ch = 0; ch < 3; ++ch) {
      const float g = drgb[ch];
      for (int k = 0; k < (drgb_out ? 1 : 16); ++k) G(OMFS_P_SH + 3 * k + ch, k < ncoef ? basis[k] * g : 0.f);
      if (cam.sh_degree > 0 && g != 0.f) {
        auto Sh = [&](int k) { return P(OMFS_P_SH + 3 * k + ch); };
        float gx = -C1 * Sh(3), gy = -C1 * Sh(1), gz = C1 * Sh(2);
        if (cam.sh_degree > 1) {
          const float s4 = Sh(4), s5 = Sh(5), s6 = Sh(6), s7 = Sh(7), s8 = Sh(8);
          gx += C2[0] * y * s4 + C2[2] * (-2.f * x) * s6 + C2[3] * z * s7 + C2[4] * 2.f * x * s8;
          gy += C2[0] * x * s4 + C2[1] * z * s5 + C2[2] * (-2.f * y) * s6 + C2[4] * (-2.f * y) * s8;
          gz += C2[1] * y * s5 + C2[2] * 4.f * z * s6 + C2[3] * x * s7;
          if (cam.sh_degree > 2) {
            const float s9 = Sh(9), s10 = Sh(10), s11 = Sh(11), s12 = Sh(12), s13 = Sh(13), s14 = Sh(14), s15 = Sh(15);
            gx += C3[0] * 6.f * xy * s9 + C3[1] * yzp * s10 + C3[2] * (-2.f * xy) * s11 + C3[3] * (-6.f * xzp) * s12 +
                  C3[4] * (4.f * zz - 3.f * xx - yy) * s13 + C3[5] * 2.f * xzp * s14 + C3[6] * (3.f * xx - 3.f * yy) * s15;
            gy += C3[0] * (3.f * xx - 3.f * yy) * s9 + C3[1] * xzp * s10 + C3[2] * (4.f * zz - xx - 3.f * yy) * s11 +
                  C3[3] * (-6.f * yzp) * s12 + C3[4] * (-2.f * xy) * s13 + C3[5] * (-2.f * yzp) * s14 + C3[6] * (-6.f * xy) * s15;
            gz += C3[1] * xy * s10 + C3[2] * 8.f * yzp * s11 + C3[3] * (6.f * zz - 3.f * xx - 3.f * yy) * s12 +
                  C3[4] * 8.f * xzp * s13 + C3[5] * (xx - yy) * s14;
          }
        }
        ddx += gx * g; ddy += gy * g; ddz += gz * g;
      }
    }
    // through d = v / |v|
    const float dd = ddx * x + ddy * y + ddz * z;
    dmu[0] += (ddx - x * dd) / vl;
    dmu[1] += (ddy - y * dd) / vl;
    dmu[2] += (ddz - z * dd) / vl;
  }

  // ---- regularisers (means over the visible Gaussians of this view)
  const float inv_nvis = 1.f / (float)max(1u, n_visible[0]);
  float dl[3], dls[3];
  // mean: l = ...;  dL/dl = sf * Rf^T dmu
  for (int k = 0; k < 3; ++k) dl[k] = sf * (Rf[k] * dmu[0] + Rf[3 + k] * dmu[1] + Rf[6 + k] * dmu[2]);
  if (reg.lambda_xyz != 0.f) {
    const float ln = sqrtf(l[0] * l[0] + l[1] * l[1] + l[2] * l[2]);
    if (ln > reg.thr_xyz) {
      const float k = reg.lambda_xyz * inv_nvis / ln;
      dl[0] += k * l[0]; dl[1] += k * l[1]; dl[2] += k * l[2];
    }
  }
  // scale: dL/ds_k = sum_r dM[r][k] Rw[r][k];  s = exp(ls) sf
  float dsk[3];   // dL/ds_k, s_k = exp(ls_k) sf
  for (int k = 0; k < 3; ++k) {
    dsk[k] = dM[k] * Rw[k] + dM[3 + k] * Rw[3 + k] + dM[6 + k] * Rw[6 + k];
    dls[k] = dsk[k] * s[k];
  }
  if (dface) {
    // gradient w.r.t. the parent triangle's frame record (FLAME fine-tuning): mu = sf Rf l + cf, M = Rf (Q diag(s)),
    // s = exp(ls) sf.  One 64-byte record per GAUSSIAN (no atomics); omfs_face_frames_bwd sums them per triangle.
    float rec[16];
    float dsf = dsk[0] * es[0] + dsk[1] * es[1] + dsk[2] * es[2];
    for (int r = 0; r < 3; ++r) {
      dsf = fma_(dmu[r], dot3_(Rf[r * 3], Rf[r * 3 + 1], Rf[r * 3 + 2], l[0], l[1], l[2]), dsf);
      for (int cc = 0; cc < 3; ++cc) {
        float gR = sf * dmu[r] * l[cc];
        for (int k = 0; k < 3; ++k) gR = fma_(dM[r * 3 + k], Q[cc * 3 + k] * s[k], gR);
        rec[r * 3 + cc] = gR;
      }
      rec[9 + r] = dmu[r];
    }
    rec[12] = dsf; rec[13] = rec[14] = rec[15] = 0.f;
    float4* o = reinterpret_cast<float4*>(dface) + (size_t)i * 4;
    for (int q = 0; q < 4; ++q) o[q] = make_float4(rec[q * 4], rec[q * 4 + 1], rec[q * 4 + 2], rec[q * 4 + 3]);
  }
  if (reg.lambda_scale != 0.f) {
    const float u0 = fmaxf(es[0] - reg.thr_scale, 0.f), u1 = fmaxf(es[1] - reg.thr_scale, 0.f), u2 = fmaxf(es[2] - reg.thr_scale, 0.f);
    const float un = sqrtf(u0 * u0 + u1 * u1 + u2 * u2);
    if (un > 0.f) {
      const float k = reg.lambda_scale * inv_nvis / un;
      dls[0] += k * u0 * es[0]; dls[1] += k * u1 * es[1]; dls[2] += k * u2 * es[2];
    }
  }
  // rotation: dRw = dM diag(s);  dQ = Rf^T dRw
  float dQ[9];
  for (int r = 0; r < 3; ++r)
    for (int k = 0; k < 3; ++k)
      dQ[r * 3 + k] = Rf[r] * dM[k] * s[k] + Rf[3 + r] * dM[3 + k] * s[k] + Rf[6 + r] * dM[6 + k] * s[k];
  const float dqw = 2.f * (-qz * dQ[1] + qy * dQ[2] + qz * dQ[3] - qx * dQ[5] - qy * dQ[6] + qx * dQ[7]);
  const float dqx = 2.f * (qy * dQ[1] + qz * dQ[2] + qy * dQ[3] - 2.f * qx * dQ[4] - qw * dQ[5] + qz * dQ[6] + qw * dQ[7] - 2.f * qx * dQ[8]);
  const float dqy = 2.f * (-2.f * qy * dQ[0] + qx * dQ[1] + qw * dQ[2] + qx * dQ[3] + qz * dQ[5] - qw * dQ[6] + qz * dQ[7] - 2.f * qy * dQ[8]);
  const float dqz = 2.f * (-2.f * qz * dQ[0] - qw * dQ[1] + qx * dQ[2] + qw * dQ[3] - 2.f * qz * dQ[4] + qy * dQ[5] + qx * dQ[6] + qy * dQ[7]);
  const float qd = dqw * qw + dqx * qx + dqy * qy + dqz * qz;

  G(OMFS_P_XYZ + 0, dl[0]); G(OMFS_P_XYZ + 1, dl[1]); G(OMFS_P_XYZ + 2, dl[2]);
  G(OMFS_P_SCALE + 0, dls[0]); G(OMFS_P_SCALE + 1, dls[1]); G(OMFS_P_SCALE + 2, dls[2]);
  G(OMFS_P_ROT + 0, (dqw - qw * qd) / qn); G(OMFS_P_ROT + 1, (dqx - qx * qd) / qn);
  G(OMFS_P_ROT + 2, (dqy - qy * qd) / qn); G(OMFS_P_ROT + 3, (dqz - qz * qd) / qn);
  const float o = 1.f / (1.f + exp_exact(-P(OMFS_P_OPACITY)));
  G(OMFS_P_OPACITY, dop * o * (1.f - o));
}

// dL/dcolour of this view per Gaussian, straight from the 2-D splat gradient records (zero for clamped channels and
// invisible Gaussians): lets the data-parallel all-gather start before project_bwd runs.
__global__ __launch_bounds__(256) void extract_drgb_kernel(int n, int n_pad, const float4* __restrict__ g2,
                                                           const float4* __restrict__ dsplat, float* __restrict__ drgb_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t rbits = __float_as_uint(g2[RI(i)].z);
  float d[3] = {0.f, 0.f, 0.f};
  if ((rbits & 0xFFFFFu) != 0u) {
    const float4 d1 = dsplat[(size_t)i * 4 + 1], d2 = dsplat[(size_t)i * 4 + 2];
    const uint32_t clampbits = rbits >> 28;
    d[0] = (clampbits & 1u) ? 0.f : d1.z;
    d[1] = (clampbits & 2u) ? 0.f : d1.w;
    d[2] = (clampbits & 4u) ? 0.f : d2.x;
  }
  for (int ch = 0; ch < 3; ++ch) drgb_out[(size_t)ch * n_pad + i] = d[ch];
}

struct ViewSetK {
  int n_views;
  int view[16];
};

// Sum over the views of Y_k(dir_w) * dL/dcolour_w for ALL 16 SH coefficients (planes 11..58; round 5: degree 0 too -- its basis is
// the constant C0, so the three SH-dc planes need not travel in the all-reduce either: 11 planes instead of 14 on the links): one
// lane per Gaussian, the mean is re-posed with each view's triangle frame (64-byte gather, L2) exactly as project_bwd does.
__global__ __launch_bounds__(256) void sh_rest_grads_kernel(int n, int n_pad, const float* __restrict__ params,
                                                            const int32_t* __restrict__ binding, const float* __restrict__ face_xf_all,
                                                            int n_faces, const float* __restrict__ cam_pos_table, ViewSetK vs,
                                                            const float* __restrict__ drgb_all, int sh_degree, float* __restrict__ grads) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float l[3] = {params[(size_t)(OMFS_P_XYZ + 0) * n_pad + i], params[(size_t)(OMFS_P_XYZ + 1) * n_pad + i],
                      params[(size_t)(OMFS_P_XYZ + 2) * n_pad + i]};
  const int face = binding[i];
  float acc[15][3], acc0[3] = {0.f, 0.f, 0.f};
  for (int k = 0; k < 15; ++k) acc[k][0] = acc[k][1] = acc[k][2] = 0.f;
  constexpr float C0 = 0.28209479177387814f, C1 = 0.4886025119029199f;
  constexpr float C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f, 0.5462742152960396f};
  constexpr float C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                           -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};
  const int ncoef = (sh_degree + 1) * (sh_degree + 1);
  for (int w = 0; w < vs.n_views; ++w) {
    const float* dr = drgb_all + (size_t)w * 3 * n_pad;
    const float g0 = dr[i], g1 = dr[(size_t)n_pad + i], g2 = dr[(size_t)2 * n_pad + i];
    if (g0 == 0.f && g1 == 0.f && g2 == 0.f) continue;
    acc0[0] = fma_(C0, g0, acc0[0]); acc0[1] = fma_(C0, g1, acc0[1]); acc0[2] = fma_(C0, g2, acc0[2]);   // degree 0: project_bwd's C0 * dL/dcolour
    const float4* fr = reinterpret_cast<const float4*>(face_xf_all) + ((size_t)w * n_faces + face) * 4;
    const float4 f0 = fr[0], f1 = fr[1], f2 = fr[2], f3 = fr[3];
    const float Rf[9] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w, f2.x};
    const float cf[3] = {f2.y, f2.z, f2.w};
    const float sf = f3.x;
    float mu[3];
    for (int r = 0; r < 3; ++r) mu[r] = fma_(dot3_(Rf[r * 3], Rf[r * 3 + 1], Rf[r * 3 + 2], l[0], l[1], l[2]), sf, cf[r]);
    const float* cp = cam_pos_table + (size_t)vs.view[w] * 3;
    const float vx = mu[0] - cp[0], vy = mu[1] - cp[1], vz = mu[2] - cp[2];
    const float vl = sqrtf(fmaxf(vx * vx + vy * vy + vz * vz, 1e-20f));
    const float x = vx / vl, y = vy / vl, z = vz / vl;
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yzp = y * z, xzp = x * z;
    float basis[15];
    basis[0] = -C1 * y; basis[1] = C1 * z; basis[2] = -C1 * x;
    basis[3] = C2[0] * xy; basis[4] = C2[1] * yzp; basis[5] = C2[2] * (2.f * zz - xx - yy); basis[6] = C2[3] * xzp; basis[7] = C2[4] * (xx - yy);
    basis[8] = C3[0] * y * (3.f * xx - yy); basis[9] = C3[1] * xy * z; basis[10] = C3[2] * y * (4.f * zz - xx - yy);
    basis[11] = C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy); basis[12] = C3[4] * x * (4.f * zz - xx - yy);
    basis[13] = C3[5] * z * (xx - yy); basis[14] = C3[6] * x * (xx - 3.f * yy);
    for (int k = 0; k < 15; ++k) {
      const float b = (k + 1) < ncoef ? basis[k] : 0.f;
      acc[k][0] = fma_(b, g0, acc[k][0]); acc[k][1] = fma_(b, g1, acc[k][1]); acc[k][2] = fma_(b, g2, acc[k][2]);
    }
  }
  for (int c = 0; c < 3; ++c) grads[(size_t)(OMFS_P_SH + c) * n_pad + i] = acc0[c];
  for (int k = 0; k < 15; ++k)
    for (int c = 0; c < 3; ++c) grads[(size_t)(OMFS_P_SH + 3 * (k + 1) + c) * n_pad + i] = acc[k][c];
}

}  // namespace omfs

using namespace omfs;

extern "C" int omfs_extract_drgb(const omfs_raster_buffers* rb, const float* dsplat, int n, int n_pad, float* drgb_out, void* stream) {
  OMFS_REQUIRE(rb && rb->g2 && dsplat && drgb_out && n > 0 && n_pad >= n, "args");
  hipLaunchKernelGGL(extract_drgb_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, n, n_pad, (const float4*)rb->g2,
                     (const float4*)dsplat, drgb_out);
  OMFS_CHECK_HIP(hipGetLastError());
  return OMFS_OK;
}

extern "C" int omfs_sh_rest_grads(const omfs_gaussians* g, const float* face_xf_all, int n_faces, const float* cam_pos_table,
                                  const omfs_view_set* views, const float* drgb_all, int sh_degree, float* grads, void* stream) {
  OMFS_REQUIRE(g && face_xf_all && cam_pos_table && views && drgb_all && grads, "null pointer");
  OMFS_REQUIRE(g->n > 0 && g->n_pad >= g->n && g->params && g->binding && n_faces > 0, "buffers");
  OMFS_REQUIRE(views->n_views >= 1 && views->n_views <= 16 && sh_degree >= 0 && sh_degree <= 3, "views");
  ViewSetK vs;
  vs.n_views = views->n_views;
  for (int w = 0; w < 16; ++w) vs.view[w] = w < views->n_views ? views->view[w] : 0;
  hipLaunchKernelGGL(sh_rest_grads_kernel, dim3(cdiv(g->n, 256)), dim3(256), 0, (hipStream_t)stream, g->n, g->n_pad, g->params,
                     g->binding, face_xf_all, n_faces, cam_pos_table, vs, drgb_all, sh_degree, grads);
  OMFS_CHECK_HIP(hipGetLastError());
  return OMFS_OK;
}

extern "C" int omfs_project_bwd(const omfs_gaussians* g, const float* face_xf, const omfs_camera* cam,
                                const omfs_raster_buffers* rb, const omfs_grad_buffers* gb,
                                const omfs_reg_params* reg, void* stream) {
  OMFS_REQUIRE(g && face_xf && cam && rb && gb && reg, "null pointer");
  OMFS_REQUIRE(g->n > 0 && g->n_pad >= g->n && g->params && g->binding && rb->g0 && rb->g1 && rb->g2 && gb->dsplat && gb->grads && reg->n_visible, "buffers");
  ProjCamB pc;
  for (int i = 0; i < 12; ++i) pc.view[i] = cam->view[i];
  for (int i = 0; i < 3; ++i) pc.cam_pos[i] = cam->cam_pos[i];
  pc.fx = cam->fx; pc.fy = cam->fy; pc.limx = cam->limx; pc.limy = cam->limy; pc.sh_degree = cam->sh_degree;
  RegK rk{reg->lambda_xyz, reg->thr_xyz, reg->lambda_scale, reg->thr_scale};
  hipLaunchKernelGGL(project_bwd_kernel, dim3(cdiv(g->n, 256)), dim3(256), 0, (hipStream_t)stream, g->n, g->n_pad,
                     g->params, g->binding, face_xf, pc, (const float4*)rb->g0, (const float4*)rb->g1, (const float4*)rb->g2, (float4*)gb->dsplat, rk,
                     reg->n_visible, gb->grads, gb->densify_stats, 0.5f * (float)cam->width, 0.5f * (float)cam->height, gb->dface, gb->drgb_out, gb->drgb_out ? gb->dir_out : nullptr);
  OMFS_CHECK_HIP(hipGetLastError());
  return OMFS_OK;
}

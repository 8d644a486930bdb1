// SimpleFLAME landmark model of the reference's flame_fitter, forward and backward.
//
// Reference: 02_Visual_Engine/flame_fitter.py:154-197 (forward) and the autograd pass of the fit
// loop :377-413.  The reference evaluates all V vertices and then mixes three of them per
// landmark; the mix is linear, so the host folds the barycentric weights into a per-landmark
// basis once (lmk_template [L][3], lmk_basis [L][3][K] with K = n_shape + n_expr, lmk_lower [L])
// and the kernels work on L landmarks instead of V vertices (68 vs 5023).  Same math, different
// summation order: parity with the reference is at fp32 tolerance, not bit level.
//
//   p      = lmk_template + lmk_basis . [shape; expr];  p.y -= jaw[0] * 0.15 * lmk_lower
//   out    = R(rotation) p + translation,  R = I + sin(t) K + (1-cos(t)) K^2, axis = aa/(|aa|+1e-8)
#include "common.hpp"

namespace omfs {

__device__ __forceinline__ void rodrigues(const float* aa, float* R, float* Kout, float& theta) {
  theta = sqrtf(aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2]);
  const float inv = 1.f / (theta + 1e-8f);
  const float n0 = aa[0] * inv, n1 = aa[1] * inv, n2 = aa[2] * inv;
  const float K[9] = {0.f, -n2, n1, n2, 0.f, -n0, -n1, n0, 0.f};
  const float s = sinf(theta), c1 = 1.f - cosf(theta);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      const float k2 = K[i * 3] * K[j] + K[i * 3 + 1] * K[3 + j] + K[i * 3 + 2] * K[6 + j];
      R[i * 3 + j] = (i == j ? 1.f : 0.f) + s * K[i * 3 + j] + c1 * k2;
    }
  if (Kout) for (int i = 0; i < 9; ++i) Kout[i] = K[i];
}

// one thread per (frame, landmark); writes landmarks [B][L][3] and the pre-rotation points p [B][L][3]
__global__ void simpleflame_fwd_kernel(int B, int L, int K, int n_shape, const float* __restrict__ lmk_template,
                                       const float* __restrict__ lmk_basis, const float* __restrict__ lmk_lower,
                                       const float* __restrict__ shape, const float* __restrict__ expr,
                                       const float* __restrict__ rotation, const float* __restrict__ jaw,
                                       const float* __restrict__ translation, float* __restrict__ out, float* __restrict__ p_out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= B * L) return;
  const int b = t / L, l = t % L;
  const int n_expr = K - n_shape;
  float p[3];
  for (int c = 0; c < 3; ++c) {
    const float* row = lmk_basis + ((size_t)l * 3 + c) * K;
    float acc = lmk_template[l * 3 + c];
    for (int k = 0; k < n_shape; ++k) acc = fma_(row[k], shape[(size_t)b * n_shape + k], acc);
    for (int k = 0; k < n_expr; ++k) acc = fma_(row[n_shape + k], expr[(size_t)b * n_expr + k], acc);
    p[c] = acc;
  }
  p[1] -= jaw[b * 3] * 0.15f * lmk_lower[l];
  float R[9], th;
  rodrigues(rotation + b * 3, R, nullptr, th);
  for (int i = 0; i < 3; ++i) {
    out[(size_t)t * 3 + i] = R[i * 3] * p[0] + R[i * 3 + 1] * p[1] + R[i * 3 + 2] * p[2] + translation[b * 3 + i];
    p_out[(size_t)t * 3 + i] = p[i];
  }
}

// one thread per frame: pose gradients + g = R^T dL/dout per landmark (for the basis products)
__global__ void simpleflame_bwd_pose_kernel(int B, int L, const float* __restrict__ lmk_lower, const float* __restrict__ rotation,
                                            const float* __restrict__ p_in, const float* __restrict__ dout,
                                            float* __restrict__ g_out, float* __restrict__ drot, float* __restrict__ djaw,
                                            float* __restrict__ dtrans) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float R[9], Km[9], th;
  const float aa[3] = {rotation[b * 3], rotation[b * 3 + 1], rotation[b * 3 + 2]};
  rodrigues(aa, R, Km, th);
  float G[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, dt[3] = {0, 0, 0}, dj = 0.f;
  for (int l = 0; l < L; ++l) {
    const float* d = dout + ((size_t)b * L + l) * 3;
    const float* p = p_in + ((size_t)b * L + l) * 3;
    float g[3];
    for (int j = 0; j < 3; ++j) g[j] = R[j] * d[0] + R[3 + j] * d[1] + R[6 + j] * d[2];
    for (int i = 0; i < 3; ++i) {
      dt[i] += d[i];
      for (int j = 0; j < 3; ++j) G[i * 3 + j] += d[i] * p[j];
      g_out[((size_t)b * L + l) * 3 + i] = g[i];
    }
    dj += -0.15f * lmk_lower[l] * g[1];
  }
  // Rodrigues backward
  const float s = sinf(th), c = cosf(th);
  float K2[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) K2[i * 3 + j] = Km[i * 3] * Km[j] + Km[i * 3 + 1] * Km[3 + j] + Km[i * 3 + 2] * Km[6 + j];
  float dth = 0.f;
  for (int i = 0; i < 9; ++i) dth += G[i] * (c * Km[i] + s * K2[i]);
  float dK[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      float gkT = 0.f, kTg = 0.f;   // (G K^T)_ij = sum_m G_im K_jm ; (K^T G)_ij = sum_m K_mi G_mj
      for (int m = 0; m < 3; ++m) { gkT += G[i * 3 + m] * Km[j * 3 + m]; kTg += Km[m * 3 + i] * G[m * 3 + j]; }
      dK[i * 3 + j] = s * G[i * 3 + j] + (1.f - c) * (gkT + kTg);
    }
  const float dn[3] = {dK[7] - dK[5], dK[2] - dK[6], dK[3] - dK[1]};
  const float inv = 1.f / (th + 1e-8f);
  const float dot = dn[0] * aa[0] + dn[1] * aa[1] + dn[2] * aa[2];
  for (int i = 0; i < 3; ++i) {
    const float unit = th > 0.f ? aa[i] / th : 0.f;   // d|aa|/daa, 0 at the origin (torch.norm's subgradient)
    drot[b * 3 + i] = dn[i] * inv - dot * inv * inv * unit + dth * unit;
    dtrans[b * 3 + i] = dt[i];
  }
  djaw[b * 3] = dj; djaw[b * 3 + 1] = 0.f; djaw[b * 3 + 2] = 0.f;
}

// one thread per (frame, coefficient): dcoef[b][k] = sum_l sum_c lmk_basis[l][c][k] * g[b][l][c]
__global__ void simpleflame_bwd_coef_kernel(int B, int L, int K, int n_shape, const float* __restrict__ lmk_basis,
                                            const float* __restrict__ g, float* __restrict__ dshape, float* __restrict__ dexpr) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= B * K) return;
  const int b = t / K, k = t % K;
  float acc = 0.f;
  for (int l = 0; l < L; ++l) {
    const float* gg = g + ((size_t)b * L + l) * 3;
    const float* row = lmk_basis + (size_t)l * 3 * K + k;
    acc = fma_(row[0], gg[0], acc);
    acc = fma_(row[K], gg[1], acc);
    acc = fma_(row[2 * K], gg[2], acc);
  }
  if (k < n_shape) dshape[(size_t)b * n_shape + k] = acc;
  else dexpr[(size_t)b * (K - n_shape) + (k - n_shape)] = acc;
}


// ---- the fit loop of fit_flame_to_landmarks (flame_fitter.py:377-413) without autograd: loss, its gradient and the
// regularisers in closed form.
__global__ void fit_broadcast_shape_kernel(int B, int S, const float* __restrict__ shape, float* __restrict__ shape_T) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B * S) shape_T[i] = shape[i % S];
}

// pseudo-perspective x/(-z+1e-8), y/(-z+1e-8) against the 2-D targets (:385-392): one thread per (frame, landmark);
// landmarks beyond n_use get a zero gradient.  loss_out += sum w (proj - target)^2 / denom.
__global__ void fit_loss_kernel(int B, int L, int n_use, const float* __restrict__ l3, const float* __restrict__ target,
                                const float* __restrict__ valid, float inv_denom, float* __restrict__ dl3,
                                float* __restrict__ loss_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  float term = 0.f;
  if (i < B * L) {
    const int b = i / L, l = i - b * L;
    float gx = 0.f, gy = 0.f, gz = 0.f;
    if (l < n_use) {
      const float x = l3[(size_t)i * 3], y = l3[(size_t)i * 3 + 1], z = l3[(size_t)i * 3 + 2];
      const float depth = -z + 1e-8f, w = valid[b];
      const float rx = x / depth - target[((size_t)b * n_use + l) * 2], ry = y / depth - target[((size_t)b * n_use + l) * 2 + 1];
      term = w * (rx * rx + ry * ry) * inv_denom;
      const float c = 2.f * w * inv_denom / depth;
      gx = c * rx; gy = c * ry;
      gz = c * (rx * x + ry * y) / depth;
    }
    dl3[(size_t)i * 3] = gx; dl3[(size_t)i * 3 + 1] = gy; dl3[(size_t)i * 3 + 2] = gz;
  }
  term = wave_sum_to_lane63(term);
  if ((threadIdx.x & 63) == 63 && term != 0.f) atomicAdd(loss_out, term);
}

// Sums the per-frame shape gradients, adds the L2 regularisers (1e-3 mean shape^2, 1e-4 mean expr^2, 1e-3 mean jaw^2,
// :395-397) and the temporal smoothness terms (1e-3 mean (x[t+1]-x[t])^2 on expr / jaw / rotation / translation,
// :400-404) to the gradients in place, and their values to loss_out.  One thread per parameter element:
// [0,S) shape, then expr [B][E], rotation, jaw, translation [B][3].
__global__ void fit_finish_grads_kernel(int B, int S, int E, const float* __restrict__ shape, const float* __restrict__ expr,
                                        const float* __restrict__ rot, const float* __restrict__ jaw,
                                        const float* __restrict__ trans, const float* __restrict__ dshape_T,
                                        float* __restrict__ dshape, float* __restrict__ dexpr, float* __restrict__ drot,
                                        float* __restrict__ djaw, float* __restrict__ dtrans, float* __restrict__ loss_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int n_expr = B * E, n3 = B * 3;
  float term = 0.f;
  auto smooth = [&](const float* x, int D, int j, float& g) {     // element j of x [B][D]
    if (B < 2) return;
    const int t = j / D;
    const float c = 1e-3f / (float)((B - 1) * D);
    float d = 0.f;
    if (t > 0) d += x[j] - x[j - D];
    if (t < B - 1) { const float f = x[j + D] - x[j]; d -= f; term += c * f * f; }
    g += 2.f * c * d;
  };
  if (i < S) {
    float a = 0.f;
    int b = 0;
    for (; b + 8 <= B; b += 8) {          // eight loads in flight: one dependent load per frame is 70 us at 300 frames
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = dshape_T[(size_t)(b + k) * S + i];
#pragma unroll
      for (int k = 0; k < 8; ++k) a += v[k];
    }
    for (; b < B; ++b) a += dshape_T[(size_t)b * S + i];
    const float v = shape[i];
    dshape[i] = a + 2e-3f * v / (float)S;
    term += 1e-3f * v * v / (float)S;
  } else if (i < S + n_expr) {
    const int j = i - S;
    float g = dexpr[j];
    const float v = expr[j];
    g += 2e-4f * v / (float)n_expr;
    term += 1e-4f * v * v / (float)n_expr;
    smooth(expr, E, j, g);
    dexpr[j] = g;
  } else if (i < S + n_expr + n3) {
    const int j = i - S - n_expr;
    float g = drot[j];
    smooth(rot, 3, j, g);
    drot[j] = g;
  } else if (i < S + n_expr + 2 * n3) {
    const int j = i - S - n_expr - n3;
    float g = djaw[j];
    const float v = jaw[j];
    g += 2e-3f * v / (float)n3;
    term += 1e-3f * v * v / (float)n3;
    smooth(jaw, 3, j, g);
    djaw[j] = g;
  } else if (i < S + n_expr + 3 * n3) {
    const int j = i - S - n_expr - 2 * n3;
    float g = dtrans[j];
    smooth(trans, 3, j, g);
    dtrans[j] = g;
  }
  term = wave_sum_to_lane63(term);
  if ((threadIdx.x & 63) == 63 && term != 0.f) atomicAdd(loss_out, term);
}

}  // namespace omfs

using namespace omfs;

extern "C" int omfs_simpleflame_fwd(const omfs_simpleflame* m, const float* shape, const float* expr, const float* rotation,
                                    const float* jaw, const float* translation, int n_frames, float* landmarks,
                                    float* p_scratch, void* stream) {
  OMFS_REQUIRE(m && shape && expr && rotation && jaw && translation && landmarks && p_scratch, "null pointer");
  OMFS_REQUIRE(n_frames > 0 && m->n_landmarks > 0 && m->n_shape >= 0 && m->n_expr >= 0, "shape");
  const int K = m->n_shape + m->n_expr, tot = n_frames * m->n_landmarks;
  hipLaunchKernelGGL(simpleflame_fwd_kernel, dim3(cdiv(tot, 128)), dim3(128), 0, (hipStream_t)stream, n_frames, m->n_landmarks, K,
                     m->n_shape, m->lmk_template, m->lmk_basis, m->lmk_lower, shape, expr, rotation, jaw, translation, landmarks, p_scratch);
  OMFS_CHECK_HIP(hipGetLastError());
  return OMFS_OK;
}

extern "C" int omfs_simpleflame_bwd(const omfs_simpleflame* m, const float* rotation, const float* p_scratch,
                                    const float* dlandmarks, int n_frames, float* g_scratch, float* dshape, float* dexpr,
                                    float* drotation, float* djaw, float* dtranslation, void* stream) {
  OMFS_REQUIRE(m && rotation && p_scratch && dlandmarks && g_scratch && dshape && dexpr && drotation && djaw && dtranslation, "null pointer");
  OMFS_REQUIRE(n_frames > 0, "shape");
  const int K = m->n_shape + m->n_expr;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(simpleflame_bwd_pose_kernel, dim3(cdiv(n_frames, 64)), dim3(64), 0, s, n_frames, m->n_landmarks, m->lmk_lower,
                     rotation, p_scratch, dlandmarks, g_scratch, drotation, djaw, dtranslation);
  OMFS_CHECK_HIP(hipGetLastError());
  hipLaunchKernelGGL(simpleflame_bwd_coef_kernel, dim3(cdiv(n_frames * K, 128)), dim3(128), 0, s, n_frames, m->n_landmarks, K,
                     m->n_shape, m->lmk_basis, g_scratch, dshape, dexpr);
  OMFS_CHECK_HIP(hipGetLastError());
  return OMFS_OK;
}

extern "C" size_t omfs_flame_fit_scratch_floats(const omfs_simpleflame* m, int n_frames) {
  if (!m || n_frames <= 0) return 0;
  const size_t B = (size_t)n_frames, L3 = (size_t)m->n_landmarks * 3;
  return 2 * B * m->n_shape + 4 * B * L3 + B * m->n_expr + 9 * B + m->n_shape;
}

extern "C" int omfs_flame_fit_step(const omfs_simpleflame* m, const omfs_flame_fit* f, void* stream) {
  OMFS_REQUIRE(m && f && f->target && f->valid && f->shape && f->expr && f->rotation && f->jaw && f->translation && f->scratch &&
                   f->loss_out, "null pointer");
  for (int k = 0; k < 5; ++k) OMFS_REQUIRE(f->m[k] && f->v[k], "Adam moments");
  const int B = f->n_frames, L = m->n_landmarks, S = m->n_shape, E = m->n_expr;
  OMFS_REQUIRE(B > 0 && f->n_use > 0 && f->n_use <= L && f->step > 0, "shape");
  hipStream_t s = (hipStream_t)stream;
  float* shape_T = f->scratch;
  float* l3 = shape_T + (size_t)B * S;
  float* p = l3 + (size_t)B * L * 3;
  float* dl3 = p + (size_t)B * L * 3;
  float* g = dl3 + (size_t)B * L * 3;
  float* dshape_T = g + (size_t)B * L * 3;
  float* dexpr = dshape_T + (size_t)B * S;
  float* drot = dexpr + (size_t)B * E;
  float* djaw = drot + (size_t)B * 3;
  float* dtrans = djaw + (size_t)B * 3;
  float* dshape = dtrans + (size_t)B * 3;
  OMFS_CHECK_HIP(hipMemsetAsync(f->loss_out, 0, sizeof(float), s));
  hipLaunchKernelGGL(fit_broadcast_shape_kernel, dim3(cdiv(B * S, 256)), dim3(256), 0, s, B, S, f->shape, shape_T);
  OMFS_CHECK_HIP(hipGetLastError());
  if (int rc = omfs_simpleflame_fwd(m, shape_T, f->expr, f->rotation, f->jaw, f->translation, B, l3, p, stream)) return rc;
  hipLaunchKernelGGL(fit_loss_kernel, dim3(cdiv(B * L, 256)), dim3(256), 0, s, B, L, f->n_use, l3, f->target, f->valid, f->inv_denom,
                     dl3, f->loss_out);
  OMFS_CHECK_HIP(hipGetLastError());
  if (int rc = omfs_simpleflame_bwd(m, f->rotation, p, dl3, B, g, dshape_T, dexpr, drot, djaw, dtrans, stream)) return rc;
  const int n_el = S + B * E + 9 * B;
  hipLaunchKernelGGL(fit_finish_grads_kernel, dim3(cdiv(n_el, 256)), dim3(256), 0, s, B, S, E, f->shape, f->expr, f->rotation, f->jaw,
                     f->translation, dshape_T, dshape, dexpr, drot, djaw, dtrans, f->loss_out);
  OMFS_CHECK_HIP(hipGetLastError());
  float* params[5] = {f->shape, f->expr, f->rotation, f->jaw, f->translation};
  const float* grads[5] = {dshape, dexpr, drot, djaw, dtrans};
  const int count[5] = {S, B * E, B * 3, B * 3, B * 3};
  for (int k = 0; k < 5; ++k)
    if (int rc = omfs_adam_flat(params[k], grads[k], f->m[k], f->v[k], count[k], f->lr[k], f->beta1, f->beta2, f->eps, f->step, 1.0f,
                                stream))
      return rc;
  return OMFS_OK;
}

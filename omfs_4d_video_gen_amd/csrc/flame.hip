// FLAME rig -> posed vertices -> per-face frames.  gfx950.
//
// Stage map (the reference's own FLAME code,
// 02_Visual_Engine/flame_fitter.py:154-197, stops at linear blendshapes and has no LBS):
//   flame_joints_kernel : J = J_static + JE.expr (15 lanes), kinematic chain -> 5 rigid transforms / frame,
//                         blendshape coefficient matrix coef[k_pad][b_pad] (expr | pose features)
//   flame_lbs_kernel    : v_posed = v_static + basis.coef on f32 MFMA (16x16x4), then linear
//                         blend skinning straight out of the accumulator registers
//   face_frames_kernel  : per triangle orthonormal frame, centroid and scale
//
// The basis product is the only GEMM-shaped work on the whole path; it is HBM/L2-bound on the
// basis (k_pad*V*3*4 bytes per call), so f32 MFMA (exact k-ordered fma chain) is used for it and
// nothing is down-converted.
#include "common.hpp"

namespace omfs {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// One 64-thread block per frame column: 15 lanes run the joint-regression chains (ascending-k fma, as
// the oracle), all lanes write the coefficient column, lane 0 walks the 5-joint kinematic chain.
__device__ __forceinline__ void rodrigues_fwd(const float* aa, float* R, float* K, float& th);

// One frame's joints by ONE 64-lane workgroup: the joint regression (15 ascending-k fma chains, as the oracle), the five rotation
// matrices (given, or -- pose != NULL -- from axis-angle poses by the formula of rodrigues_kernel, bit for bit, and stored to
// rotmats_out when that is not NULL), the blendshape coefficient column (expr, then the 36 pose features R_j - I, then zeros) and
// the 5-joint kinematic chain.  Outputs go wherever the caller points: coef_out[k * coef_stride] (global column or an LDS
// array), X_out[60] (global or LDS; written by lane 0).  Ends with a workgroup barrier: the outputs are visible to the block.
struct JointsLds {
  float sJ[15];
  float sR[45];
  float s_je[15 * 128];
  float s_e[128];
};
__device__ __forceinline__ void joints_frame(JointsLds& L, const float* __restrict__ j_static, const float* __restrict__ j_expr,
                                             const float* __restrict__ e, const float* __restrict__ rot_in, const float* __restrict__ pose,
                                             float* __restrict__ rotmats_out, int n_expr, int k_pad, float* coef_out, int coef_stride,
                                             float* X_out) {
  const int lane = threadIdx.x;
  float* sJ = L.sJ; float* sR = L.sR; float* s_je = L.s_je; float* s_e = L.s_e;
  if (pose) {
    if (lane < 5) {
      float Rj[9], Kj[9], th;
      rodrigues_fwd(pose + lane * 3, Rj, Kj, th);
      for (int i = 0; i < 9; ++i) { sR[lane * 9 + i] = Rj[i]; if (rotmats_out) rotmats_out[lane * 9 + i] = Rj[i]; }
    }
  } else if (lane < 45) {
    sR[lane] = rot_in[lane];
  }
  // the joint regressor's expression part (15 x n_expr) and the coefficients go through LDS: all 64 lanes fetch them with
  // independent coalesced loads (one memory round trip), the 15 ascending-k fma chains then run from LDS -- straight from
  // global memory every chain step was its own round trip (14 us alone, 58-81 us beside the Adam pass)
  // all loads of a pass in flight together (a plain copy loop waits for every load before its LDS store: 24 round trips
  // for the 1500 words of the regressor -- 32 us for this kernel beside the Adam pass, where a round trip takes 3-5x longer)
  const float jst = lane < 15 ? j_static[lane] : 0.f;
  const int n_je = 15 * n_expr;
  for (int k0 = lane; k0 < n_je; k0 += 24 * 64) {      // 24 x 64 = 1536 words: one pass for FLAME's 15 x 100
    float v[24];
#pragma unroll
    for (int u = 0; u < 24; ++u) { const int k = k0 + u * 64; v[u] = k < n_je ? j_expr[k] : 0.f; }
#pragma unroll
    for (int u = 0; u < 24; ++u) { const int k = k0 + u * 64; if (k < n_je) s_je[k] = v[u]; }
  }
  {
    const float e0 = lane < n_expr ? e[lane] : 0.f, e1 = lane + 64 < n_expr ? e[lane + 64] : 0.f;
    if (lane < n_expr) s_e[lane] = e0;
    if (lane + 64 < n_expr) s_e[lane + 64] = e1;
  }
  __syncthreads();
  const float* R = sR;
  if (lane < 15) {  // J[j][c] = j_static + sum_k j_expr[j*3+c][k] * e[k]
    float acc = jst;
    const float* row = s_je + lane * n_expr;
    for (int k = 0; k < n_expr; ++k) acc = fma_(row[k], s_e[k], acc);
    sJ[lane] = acc;
  }
  // coefficient column: expr, then pose features (R_j - I), j = 1..4, row-major, then zero padding
  for (int k = lane; k < k_pad; k += 64) {
    float v = 0.f;
    if (k < n_expr) v = s_e[k];
    else if (k < n_expr + 36) {
      const int i = (k - n_expr) % 9;
      v = R[9 + (k - n_expr)] - ((i == 0 || i == 4 || i == 8) ? 1.f : 0.f);
    }
    coef_out[(size_t)k * coef_stride] = v;
  }
  __syncthreads();
  if (lane == 0) {
    float J[5][3];
    for (int jc = 0; jc < 15; ++jc) J[jc / 3][jc % 3] = sJ[jc];
    // kinematic chain, parents = [-1, 0, 1, 1, 1]
    float Rw[5][9], tw[5][3];
    for (int i = 0; i < 9; ++i) Rw[0][i] = R[i];
    for (int c = 0; c < 3; ++c) tw[0][c] = J[0][c];
    for (int j = 1; j < 5; ++j) {
      int p = (j == 1) ? 0 : 1;
      const float* Rl = R + j * 9;
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
          Rw[j][r * 3 + c] = dot3_(Rw[p][r * 3 + 0], Rw[p][r * 3 + 1], Rw[p][r * 3 + 2], Rl[c], Rl[3 + c], Rl[6 + c]);
      float dx = J[j][0] - J[p][0], dy = J[j][1] - J[p][1], dz = J[j][2] - J[p][2];
      for (int r = 0; r < 3; ++r) tw[j][r] = dot3_(Rw[p][r * 3 + 0], Rw[p][r * 3 + 1], Rw[p][r * 3 + 2], dx, dy, dz) + tw[p][r];
    }
    for (int j = 0; j < 5; ++j) {
      for (int i = 0; i < 9; ++i) X_out[j * 12 + i] = Rw[j][i];
      for (int r = 0; r < 3; ++r)
        X_out[j * 12 + 9 + r] = tw[j][r] - dot3_(Rw[j][r * 3 + 0], Rw[j][r * 3 + 1], Rw[j][r * 3 + 2], J[j][0], J[j][1], J[j][2]);
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(64) void flame_joints_kernel(const float* __restrict__ j_static, const float* __restrict__ j_expr,
                                                          const float* __restrict__ expr, float* __restrict__ rotmats,
                                                          const float* __restrict__ pose,
                                                          int n_frames, int n_expr, int k_pad, int b_pad,
                                                          float* __restrict__ joint_xf, float* __restrict__ coef,
                                                          const int32_t* __restrict__ frame_index) {
  __shared__ JointsLds L;
  const int b = blockIdx.x, lane = threadIdx.x;
  if (b >= n_frames) {  // padded frame columns: zero coefficients
    for (int k = lane; k < k_pad; k += 64) coef[(size_t)k * b_pad + b] = 0.f;
    return;
  }
  const int src = frame_index ? frame_index[b] : b;     // row of the sequence arrays this batch column shows
  joints_frame(L, j_static, j_expr, expr + (size_t)src * n_expr, rotmats + (size_t)src * 45, pose ? pose + (size_t)src * 15 : nullptr,
               rotmats + (size_t)src * 45, n_expr, k_pad, coef + b, b_pad, joint_xf + (size_t)b * 60);
}

// grid = (v_pad/16 strips, b_pad/16 column blocks), block = 64 (one wave per 16 vertices x 16 frames).
// A tile (c, strip, kt): 64 lanes x float4; lane l holds basis rows k = 16*kt + 4*j + (l>>4), j=0..3,
// for vertex strip*16 + (l&15).  MFMA j of tile kt therefore consumes k = 16kt+4j .. 16kt+4j+3 in
// order: the accumulator is a single ascending-k fma chain (bit-exact vs the oracle's fmaf loop).
__global__ __launch_bounds__(64) void flame_lbs_kernel(const float* __restrict__ basis_tiled,
                                                       const float* __restrict__ v_static,
                                                       const float* __restrict__ lbs_weights,
                                                       const float* __restrict__ coef,
                                                       const float* __restrict__ joint_xf,
                                                       const float* __restrict__ translation,
                                                       const float* __restrict__ dynamic_offset, int n_verts, int v_pad,
                                                       int k_pad, int n_frames, int b_pad, float* __restrict__ verts,
                                                       float* __restrict__ v_shaped_out, const int32_t* __restrict__ frame_index) {
  const int strip = blockIdx.x, cb = blockIdx.y;
  const int lane = threadIdx.x;
  const int n_strips = v_pad / 16, n_kt = k_pad / 16;
  const int col = lane & 15, grp = lane >> 4;
  const int frame = cb * 16 + col;
  f32x4 acc[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    // C/D map of 16x16x4 f32: row = grp*4 + reg (vertex in strip), col = lane&15 (frame)
    const float* vs = v_static + (size_t)c * v_pad + strip * 16 + grp * 4;
    acc[c] = f32x4{vs[0], vs[1], vs[2], vs[3]};
  }
  // five k-tiles of operands in flight per pass (the MFMA chain itself stays in ascending k): with one tile per iteration
  // every iteration was a memory round trip of its own -- nine of them, 12 us alone and 27-31 us beside the Adam pass
  constexpr int KU = 5;
  for (int kt0 = 0; kt0 < n_kt; kt0 += KU) {
    f32x4 a[KU][3];
    float bv[KU][4];
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      const int kt = kt0 + u < n_kt ? kt0 + u : n_kt - 1;        // clamped: the surplus loads are not used
#pragma unroll
      for (int c = 0; c < 3; ++c)
        a[u][c] = *reinterpret_cast<const f32x4*>(basis_tiled + ((((size_t)c * n_strips + strip) * n_kt + kt) * 64 + lane) * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j)   // B operand: lane holds coef[k = 16kt + 4j + grp][frame col]
        bv[u][j] = coef[(size_t)(kt * 16 + j * 4 + grp) * b_pad + cb * 16 + col];
    }
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      if (kt0 + u >= n_kt) break;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][c][j], bv[u][j], acc[c], 0, 0, 0);
    }
  }
  if (frame >= n_frames) return;
  const float* X = joint_xf + (size_t)frame * 60;
  const int src = frame_index ? frame_index[frame] : frame;
  const float tx = translation[src * 3 + 0], ty = translation[src * 3 + 1], tz = translation[src * 3 + 2];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    int v = strip * 16 + grp * 4 + r;
    if (v >= n_verts) continue;
    const float* w = lbs_weights + (size_t)v * 8;
    float M[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      float m = w[0] * X[i];
#pragma unroll
      for (int j = 1; j < 5; ++j) m = fma_(w[j], X[j * 12 + i], m);
      M[i] = m;
    }
    float x = acc[0][r], y = acc[1][r], z = acc[2][r];
    if (v_shaped_out) *reinterpret_cast<float4*>(v_shaped_out + ((size_t)frame * v_pad + v) * 4) = make_float4(x, y, z, 1.f);
    float ox = dot3_(M[0], M[1], M[2], x, y, z) + M[9];
    float oy = dot3_(M[3], M[4], M[5], x, y, z) + M[10];
    float oz = dot3_(M[6], M[7], M[8], x, y, z) + M[11];
    if (dynamic_offset) {
      const float* d = dynamic_offset + ((size_t)src * n_verts + v) * 3;
      ox += d[0]; oy += d[1]; oz += d[2];
    }
    ox += tx; oy += ty; oz += tz;
    *reinterpret_cast<float4*>(verts + ((size_t)frame * v_pad + v) * 4) = make_float4(ox, oy, oz, 1.f);
  }
}

// ONE frame, joints and skinning in ONE launch (flame_joints_kernel + flame_lbs_kernel for n_frames = 1: the training step poses
// one view per iteration, and the joints launch was a 7 us kernel plus a kernel boundary in front of every skinning pass).
// Every wave (16 vertices) evaluates the frame's joints itself -- joints_frame, the code of flame_joints_kernel, so the same
// bits -- while its first basis tiles are in flight; the coefficient column and the five transforms stay in LDS (strip 0 also
// writes them to joint_xf / coef / rotmats for the backward pass and the batched path).  The MFMA B operand holds the
// coefficients in frame column 0 and zeros in the 15 padded columns, exactly what the coef matrix of a one-frame batch holds.
// The skinning epilogue runs on 16 lanes, one vertex each (the accumulators go through LDS), instead of 4 lanes x 4 vertices.
__global__ __launch_bounds__(64) void flame_pose_lbs_kernel(const float* __restrict__ basis_tiled, const float* __restrict__ v_static,
                                                            const float* __restrict__ lbs_weights, const float* __restrict__ j_static,
                                                            const float* __restrict__ j_expr, const float* __restrict__ expr,
                                                            float* __restrict__ rotmats, const float* __restrict__ pose,
                                                            const float* __restrict__ translation,
                                                            const float* __restrict__ dynamic_offset, int n_verts, int v_pad, int k_pad,
                                                            int n_expr, float* __restrict__ verts, float* __restrict__ v_shaped_out,
                                                            float* __restrict__ joint_xf, float* __restrict__ coef,
                                                            const int32_t* __restrict__ frame_index) {
  __shared__ JointsLds L;
  __shared__ float s_coef[192];       // k_pad <= 176 (n_expr <= 128)
  __shared__ float sX[60];
  __shared__ float s_acc[16][4];
  const int strip = blockIdx.x, lane = threadIdx.x;
  const int n_strips = v_pad / 16, n_kt = k_pad / 16;
  const int col = lane & 15, grp = lane >> 4;
  f32x4 acc[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float* vs = v_static + (size_t)c * v_pad + strip * 16 + grp * 4;
    acc[c] = f32x4{vs[0], vs[1], vs[2], vs[3]};
  }
  constexpr int KU = 5;
  f32x4 a[KU][3];
  auto load_tiles = [&](int kt0) {
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      const int kt = kt0 + u < n_kt ? kt0 + u : n_kt - 1;        // clamped: the surplus loads are not used
#pragma unroll
      for (int c = 0; c < 3; ++c)
        a[u][c] = *reinterpret_cast<const f32x4*>(basis_tiled + ((((size_t)c * n_strips + strip) * n_kt + kt) * 64 + lane) * 4);
    }
  };
  load_tiles(0);                      // in flight during the joints
  const int src = frame_index ? frame_index[0] : 0;
  // epilogue operands of this lane's vertex, also in flight now
  const int ve = strip * 16 + col;
  float w[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  float dyn[3] = {0.f, 0.f, 0.f};
  const bool eon = grp == 0 && ve < n_verts;
  if (eon) {
#pragma unroll
    for (int j = 0; j < 5; ++j) w[j] = lbs_weights[(size_t)ve * 8 + j];
    if (dynamic_offset) {
      const float* d = dynamic_offset + ((size_t)src * n_verts + ve) * 3;
      dyn[0] = d[0]; dyn[1] = d[1]; dyn[2] = d[2];
    }
  }
  const float tx = translation[src * 3 + 0], ty = translation[src * 3 + 1], tz = translation[src * 3 + 2];
  const bool first = strip == 0;
  joints_frame(L, j_static, j_expr, expr + (size_t)src * n_expr, rotmats + (size_t)src * 45, pose ? pose + (size_t)src * 15 : nullptr,
               first ? rotmats + (size_t)src * 45 : nullptr, n_expr, k_pad, s_coef, 1, sX);
  if (first) {
    for (int k = lane; k < k_pad; k += 64) coef[(size_t)k * 16] = s_coef[k];     // column 0 of the [k_pad][16] matrix; 1..15 stay zero
    if (lane < 60) joint_xf[lane] = sX[lane];
  }
  for (int kt0 = 0; kt0 < n_kt; kt0 += KU) {
    if (kt0 > 0) load_tiles(kt0);
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      if (kt0 + u >= n_kt) break;
      const int kt = kt0 + u;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float bv = col == 0 ? s_coef[kt * 16 + j * 4 + grp] : 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][c][j], bv, acc[c], 0, 0, 0);
      }
    }
  }
  // C/D map of 16x16x4 f32: row = grp*4 + reg (vertex in strip), col = lane&15 (frame): frame 0 lives in the lanes with col == 0
  if (col == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) { s_acc[grp * 4 + r][0] = acc[0][r]; s_acc[grp * 4 + r][1] = acc[1][r]; s_acc[grp * 4 + r][2] = acc[2][r]; }
  }
  __syncthreads();
  if (!eon) return;
  const float* X = sX;
  float M[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    float m = w[0] * X[i];
#pragma unroll
    for (int j = 1; j < 5; ++j) m = fma_(w[j], X[j * 12 + i], m);
    M[i] = m;
  }
  const float x = s_acc[col][0], y = s_acc[col][1], z = s_acc[col][2];
  if (v_shaped_out) *reinterpret_cast<float4*>(v_shaped_out + (size_t)ve * 4) = make_float4(x, y, z, 1.f);
  float ox = dot3_(M[0], M[1], M[2], x, y, z) + M[9];
  float oy = dot3_(M[3], M[4], M[5], x, y, z) + M[10];
  float oz = dot3_(M[6], M[7], M[8], x, y, z) + M[11];
  if (dynamic_offset) { ox += dyn[0]; oy += dyn[1]; oz += dyn[2]; }
  ox += tx; oy += ty; oz += tz;
  *reinterpret_cast<float4*>(verts + (size_t)ve * 4) = make_float4(ox, oy, oz, 1.f);
}

__device__ __forceinline__ void safe_normalize3(float& x, float& y, float& z) {
  float d = fmaxf(dot3_(x, y, z, x, y, z), 1e-20f);
  float l = sqrtf(d);
  x = x / l; y = y / l; z = z / l;
}

// One thread per (face, frame).
__global__ void face_frames_kernel(const float* __restrict__ verts, int v_pad, const int32_t* __restrict__ faces,
                                   int n_faces, int n_frames, float* __restrict__ face_xf) {
  int f = blockIdx.x * blockDim.x + threadIdx.x;
  int b = blockIdx.y;
  if (f >= n_faces) return;
  const float4* vb = reinterpret_cast<const float4*>(verts) + (size_t)b * v_pad;
  int i0 = faces[f * 3 + 0], i1 = faces[f * 3 + 1], i2 = faces[f * 3 + 2];
  float4 v0 = vb[i0], v1 = vb[i1], v2 = vb[i2];
  float e1x = v1.x - v0.x, e1y = v1.y - v0.y, e1z = v1.z - v0.z;
  float e2x = v2.x - v0.x, e2y = v2.y - v0.y, e2z = v2.z - v0.z;
  float a0x = e1x, a0y = e1y, a0z = e1z;
  safe_normalize3(a0x, a0y, a0z);
  // n = normalize(a0 x e2)
  float nx = fma_(a0y, e2z, -(a0z * e2y)), ny = fma_(a0z, e2x, -(a0x * e2z)), nz = fma_(a0x, e2y, -(a0y * e2x));
  safe_normalize3(nx, ny, nz);
  // a2 = -normalize(n x a0)
  float cx = fma_(ny, a0z, -(nz * a0y)), cy = fma_(nz, a0x, -(nx * a0z)), cz = fma_(nx, a0y, -(ny * a0x));
  safe_normalize3(cx, cy, cz);
  float a2x = -cx, a2y = -cy, a2z = -cz;
  float s0 = sqrtf(dot3_(e1x, e1y, e1z, e1x, e1y, e1z));
  float s1 = fabsf(dot3_(a2x, a2y, a2z, e2x, e2y, e2z));
  float scale = (s0 + s1) * 0.5f;
  const float third = 1.0f / 3.0f;
  float ccx = ((v0.x + v1.x) + v2.x) * third, ccy = ((v0.y + v1.y) + v2.y) * third, ccz = ((v0.z + v1.z) + v2.z) * third;
  float4* o = reinterpret_cast<float4*>(face_xf) + ((size_t)b * n_faces + f) * 4;
  // R row-major, columns (a0, n, a2)
  o[0] = make_float4(a0x, nx, a2x, a0y);
  o[1] = make_float4(ny, a2y, a0z, nz);
  o[2] = make_float4(a2z, ccx, ccy, ccz);
  o[3] = make_float4(scale, 0.f, 0.f, 0.f);
}

// ---- backward (FLAME fine-tuning), one frame.  SIXTEEN lanes per face (one DPP row; four faces per wave): lane q sums word q
// of the frame-gradient records project_bwd wrote for the Gaussians bound to this triangle (CSR by triangle; no atomics
// there) -- a record is one 64-byte line across the row, the sixteen Gaussian indices of a batch are fetched in one
// coalesced load and handed round by lane permutes, sixteen record loads are then in flight at once: two memory round
// trips per sixteen Gaussians instead of two per four (the chain runs beside the Adam pass, where a round trip is slow).
// Every lane then holds all thirteen sums (row permutes) and evaluates the small gradient of the frame record
// (R columns a0, n, a2; centre; scale) w.r.t. the three vertices; lanes 0..8 add one component each into dverts with a
// float atomic (a vertex belongs to ~6 faces).  The clamps of safe_normalize3 are not differentiated.
// FX (omfs_face_frames_bwd_fx): the nine corner contributions are added as 64-bit fixed-point integers (scale 2^40) -- integer
// addition is associative, the totals do not depend on the order the faces arrive in -- and converted by dverts_from_fixed_kernel.
template <bool FX>
__global__ __launch_bounds__(256) void face_frames_bwd_kernel(const float* __restrict__ verts, const int32_t* __restrict__ faces, int n_faces,
                                                              const float* __restrict__ dface, const int32_t* __restrict__ face_start,
                                                              const int32_t* __restrict__ face_gauss, float* __restrict__ dverts,
                                                              long long* __restrict__ dverts_fx) {
  const int f = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const int q = threadIdx.x & 15, lane = threadIdx.x & 63, row0 = lane & 48;
  const bool live = f < n_faces;
  const int lo = live ? face_start[f] : 0, hi = live ? face_start[f + 1] : 0;
  float acc = 0.f;
  // the triangle's own vertices: fetched now, used after the gather (two more round trips otherwise)
  const float4* vb = reinterpret_cast<const float4*>(verts);
  const int i0 = live ? faces[f * 3 + 0] : 0, i1 = live ? faces[f * 3 + 1] : 0, i2 = live ? faces[f * 3 + 2] : 0;
  const float4 v0 = vb[i0], v1 = vb[i1], v2 = vb[i2];
  // every row of the wave runs the same number of batches (lane permutes are wave-wide instructions)
  int nb = (hi - lo + 15) >> 4;
#pragma unroll
  for (int d = 16; d < 64; d <<= 1) nb = max(nb, __shfl_xor(nb, d, 64));
  for (int b = 0; b < nb; ++b) {
    const int e = lo + b * 16 + q;
    const int mine = e < hi ? face_gauss[e] : -1;
    float part[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int id = __shfl(mine, row0 + k, 64);
      part[k] = id >= 0 ? dface[(size_t)id * 16 + q] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) acc += part[k];
  }
  float g[13];
#pragma unroll
  for (int k = 0; k < 13; ++k) g[k] = __shfl(acc, row0 + k, 64);
  if (lo == hi) return;
  float da0[3] = {g[0], g[3], g[6]}, dn[3] = {g[1], g[4], g[7]}, da2[3] = {g[2], g[5], g[8]};
  const float dc[3] = {g[9], g[10], g[11]};
  const float ds = g[12];
  auto cross = [](const float* a, const float* b, float* o) {
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
  };
  auto dot = [](const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; };
  const float e1[3] = {v1.x - v0.x, v1.y - v0.y, v1.z - v0.z}, e2[3] = {v2.x - v0.x, v2.y - v0.y, v2.z - v0.z};
  const float L1 = sqrtf(fmaxf(dot(e1, e1), 1e-20f));
  const float a0[3] = {e1[0] / L1, e1[1] / L1, e1[2] / L1};
  float m[3]; cross(a0, e2, m);
  const float Ln = sqrtf(fmaxf(dot(m, m), 1e-20f));
  const float n[3] = {m[0] / Ln, m[1] / Ln, m[2] / Ln};
  float p[3]; cross(n, a0, p);
  const float Lp = sqrtf(fmaxf(dot(p, p), 1e-20f));
  const float a2[3] = {-p[0] / Lp, -p[1] / Lp, -p[2] / Lp};
  const float h = dot(a2, e2);
  // scale = (L1 + |h|) / 2
  const float dL1 = 0.5f * ds, dh = 0.5f * ds * (h > 0.f ? 1.f : (h < 0.f ? -1.f : 0.f));
  float de2[3];
  for (int k = 0; k < 3; ++k) { da2[k] += dh * e2[k]; de2[k] = dh * a2[k]; }
  // a2 = -p / |p|
  const float t2 = dot(a2, da2);
  float dp[3];
  for (int k = 0; k < 3; ++k) dp[k] = -(da2[k] - a2[k] * t2) / Lp;
  // p = n x a0
  float tmp[3];
  cross(a0, dp, tmp); for (int k = 0; k < 3; ++k) dn[k] += tmp[k];
  cross(dp, n, tmp);  for (int k = 0; k < 3; ++k) da0[k] += tmp[k];
  // n = m / |m|
  const float tn = dot(n, dn);
  float dm[3];
  for (int k = 0; k < 3; ++k) dm[k] = (dn[k] - n[k] * tn) / Ln;
  // m = a0 x e2
  cross(e2, dm, tmp); for (int k = 0; k < 3; ++k) da0[k] += tmp[k];
  cross(dm, a0, tmp); for (int k = 0; k < 3; ++k) de2[k] += tmp[k];
  // a0 = e1 / |e1|, L1 = |e1|
  const float t0 = dot(a0, da0);
  float de1[3];
  for (int k = 0; k < 3; ++k) de1[k] = (da0[k] - a0[k] * t0) / L1 + dL1 * a0[k];
  const float third = 1.0f / 3.0f;
  if (q < 9) {                      // lane q of the row: vertex q / 3, component q % 3
    const int k = q % 3, vtx = q / 3;
    const float c3 = dc[k] * third;
    const float val = vtx == 0 ? c3 - de1[k] - de2[k] : (vtx == 1 ? c3 + de1[k] : c3 + de2[k]);
    const int iv = vtx == 0 ? i0 : (vtx == 1 ? i1 : i2);
    if (FX) {
      const float sc = fminf(fmaxf(val * 1099511627776.f, -4.6e18f), 4.6e18f);     // 2^40; |.| < 2^62
      atomicAdd(reinterpret_cast<unsigned long long*>(dverts_fx) + (size_t)iv * 4 + k, (unsigned long long)(long long)__builtin_rintf(sc));
    } else {
      atomicAdd(&dverts[(size_t)iv * 4 + k], val);
    }
  }
}

__global__ void dverts_from_fixed_kernel(long long* __restrict__ fx, float* __restrict__ dverts, int n4) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4 || (i & 3) == 3) return;
  const long long v = fx[i];
  if (v == 0) return;
  fx[i] = 0;
  dverts[i] = (float)((double)v * (1.0 / 1099511627776.0));
}

// One thread per vertex: v_posed = M_v [v_shaped; 1] + ..., M_v = sum_j w_vj X_j; v_shaped [v_pad][4] was stored by
// flame_lbs.  Writes dv_shaped = M_v(3x3)^T dv and the per-wave partial sums of d X_j = w_vj dv (x) [v_shaped; 1] and
// d translation = dv (63 values, DPP wave reduction) into sums[wave][64].
__global__ __launch_bounds__(256) void flame_skin_bwd_kernel(const float* __restrict__ lbs_weights, const float* __restrict__ v_shaped,
                                                             const float* __restrict__ joint_xf, float* __restrict__ dverts,
                                                             int n_verts, float* __restrict__ dv_shaped, float* __restrict__ sums) {
  __shared__ float X[60];
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  const bool on = v < n_verts;
  float w[5] = {0.f, 0.f, 0.f, 0.f, 0.f}, dv[3] = {0.f, 0.f, 0.f}, vs[4] = {0.f, 0.f, 0.f, 1.f};
  float4 d = make_float4(0.f, 0.f, 0.f, 0.f), a = d;
  if (on) {                // every load of the thread is in flight before the barrier (one memory round trip, not two)
    for (int j = 0; j < 5; ++j) w[j] = lbs_weights[(size_t)v * 8 + j];
    d = reinterpret_cast<const float4*>(dverts)[v]; a = reinterpret_cast<const float4*>(v_shaped)[v];
  }
  if (threadIdx.x < 60) X[threadIdx.x] = joint_xf[threadIdx.x];
  __syncthreads();
  if (on) {
    reinterpret_cast<float4*>(dverts)[v] = make_float4(0.f, 0.f, 0.f, 0.f);   // consumed: the next frame's atomics start from zero
    dv[0] = d.x; dv[1] = d.y; dv[2] = d.z;
    vs[0] = a.x; vs[1] = a.y; vs[2] = a.z;
    float out[3] = {0.f, 0.f, 0.f};
    for (int j = 0; j < 5; ++j)
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) out[c] = fma_(X[j * 12 + r * 3 + c], w[j] * dv[r], out[c]);
    for (int c = 0; c < 3; ++c) dv_shaped[(size_t)v * 3 + c] = out[c];
  }
  // one row of 64 partial sums per wave (no same-address atomics); flame_front_bwd adds the rows up
  const bool last = (threadIdx.x & 63) == 63;
  float* row = sums + (size_t)(blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)) * 64;
  for (int j = 0; j < 5; ++j)
    for (int r = 0; r < 3; ++r) {
      const float g = w[j] * dv[r];
      for (int c = 0; c < 4; ++c) {
        const float t = wave_sum_to_lane63(g * vs[c]);
        if (last) row[j * 12 + (c < 3 ? r * 3 + c : 9 + r)] = t;
      }
    }
  for (int c = 0; c < 3; ++c) {
    const float t = wave_sum_to_lane63(dv[c]);
    if (last) row[60 + c] = t;
  }
  if (last) row[63] = 0.f;
}

// dcoef[k] = sum_i basis_dense[k][i] dv_shaped[i]   (one block per coefficient)
struct FrontArgs {
  const float* j_static; const float* j_expr; const float* expr; const float* pose; int n_expr; const float* partial; int n_rows;
  float* dexpr; float* dpose; float* dtrans;
  int totals;     // partial is ONE row of totals (flame_skin_gemv_kernel) instead of n_rows per-wave rows (flame_skin_bwd_kernel)
};
// LDS of the front: what its forward half leaves for its backward half
struct FrontLds {
  float sJ[15], sdJ[15], sums[64], part16[16][64], s_dcoef[256], s_pose[15];
  float s_je[15 * 128];   // joint regressor's expression part and the coefficients: one coalesced round trip
  float s_e[128];
  float sR[5][9], sK[5][9], sth[5];
};
// forward half (joint regression, the five Rodrigues maps): depends on nothing the backward pass produces, so a kernel may run it
// early, in the shadow of its own loads; called by all threads of a block, ends on a barrier
__device__ void flame_front_prepare(FrontLds& F, const FrontArgs& fa);
// backward half: totals -> chain -> axis-angle maps -> d expr; called by all threads of the block that ran flame_front_prepare
__device__ void flame_front_finish(FrontLds& F, const FrontArgs& fa, const float* dcoef);
__device__ __forceinline__ void flame_front_bwd(FrontLds& F, const FrontArgs& fa, const float* dcoef) {
  flame_front_prepare(F, fa);
  flame_front_finish(F, fa, dcoef);
}

// The block that finishes last (ticket in dcoef[gridDim.x], reset for the next call) goes on with flame_front_bwd: the
// basis^T product and the small serial front share one launch.
constexpr int GEMV_NT = 1024, GEMV_PER = 16;     // rows of up to 16384 elements (3 V = 15 429) in ONE memory round trip
__global__ __launch_bounds__(GEMV_NT) void basis_t_gemv_kernel(const float* __restrict__ basis_dense, const float* __restrict__ dv_shaped,
                                                               int row, float* __restrict__ dcoef, FrontArgs fa) {
  __shared__ float ws[GEMV_NT / 64];
  __shared__ uint32_t s_ticket;
  const float* b = basis_dense + (size_t)blockIdx.x * row;
  float acc = 0.f;
  for (int i0 = 0; i0 < row; i0 += GEMV_NT * GEMV_PER) {
    float bv[GEMV_PER], dv[GEMV_PER];
#pragma unroll
    for (int u = 0; u < GEMV_PER; ++u) {             // all loads of the pass are issued before the first use
      const int i = i0 + u * GEMV_NT + threadIdx.x;
      bv[u] = i < row ? b[i] : 0.f;
      dv[u] = i < row ? dv_shaped[i] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < GEMV_PER; ++u) acc = fma_(bv[u], dv[u], acc);
  }
  acc = wave_sum_all(acc);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < GEMV_NT / 64; ++w) t += ws[w];
    dcoef[blockIdx.x] = t;
    __threadfence();
    s_ticket = atomicAdd(reinterpret_cast<uint32_t*>(dcoef + gridDim.x), 1u);
  }
  __syncthreads();
  if (s_ticket != gridDim.x - 1) return;        // uniform over the block
  __threadfence();
  if (threadIdx.x == 0) *reinterpret_cast<uint32_t*>(dcoef + gridDim.x) = 0u;
  __shared__ FrontLds F;
  flame_front_bwd(F, fa, dcoef);
}

// Skinning backward + basis^T product + the serial front in ONE launch (flame_skin_bwd_kernel + basis_t_gemv_kernel; their
// dv_shaped round trip through memory and one kernel boundary are gone).  A workgroup owns SKG_V consecutive vertices:
//   1. lanes < SKG_V turn dL/d(posed vertex) into dL/d(blend-shaped vertex) dvs = M_v^T dv (M_v = sum_j w_vj X_j) and stage it
//      with (w, dv, v_shaped) in LDS; the dverts rows are consumed (left zeroed);
//   2. thread k < n_coef sums basis_t[3 v + c][k] * dvs[v][c] over the workgroup's 3 SKG_V columns -- basis_t is the basis
//      TRANSPOSED ([3 V][n_coef]), so the wave's loads are contiguous, every column value is an LDS broadcast and NO cross-lane
//      reduction exists; all loads of a thread are issued before the barrier that publishes dvs; one float atomic per (block, k);
//   3. thread q < 63 sums its entry of { d joint_xf [5][12], d translation [3] } over the workgroup's vertices from LDS (no wave
//      reductions: flame_skin_bwd_kernel spent 63 x 7 DPP instructions per wave on them) and adds it with one float atomic;
//   4. the workgroup that finishes last (ticket) reads the totals, zeroes them for the next call and runs flame_front_bwd.
// Float atomics on ONE cache line serialise at the memory side (~11 ns each): 161 workgroups adding into the same 136 words took
// 22 us.  The workgroups therefore add into SKG_G copies of the accumulators (workgroup b into copy b mod SKG_G: a tenth of the
// adds per line, sixteen times the lines), and the last workgroup adds the copies up while it consumes them.
#ifndef OMFS_SKG_V
#define OMFS_SKG_V 32
#endif
constexpr int SKG_V = OMFS_SKG_V, SKG_NT = 256, SKG_COLS = 3 * SKG_V, SKG_G = 16;
__global__ __launch_bounds__(SKG_NT) void flame_skin_gemv_kernel(const float* __restrict__ lbs_weights, const float* __restrict__ v_shaped,
                                                                 const float* __restrict__ joint_xf, float* __restrict__ dverts, int n_verts,
                                                                 const float* __restrict__ basis_t, int n_coef, float* __restrict__ dcoef,
                                                                 float* __restrict__ sums, FrontArgs fa) {
  __shared__ float X[60];
  __shared__ float s_w[SKG_V][5], s_dv[SKG_V][3], s_vs[SKG_V][4];
  __shared__ float s_dvs[SKG_COLS];
  __shared__ uint32_t s_ticket;
  const int tid = threadIdx.x, v0 = blockIdx.x * SKG_V;
  const int nv = min(SKG_V, n_verts - v0), ncol = 3 * nv;
  // the thread's basis column slice: independent of everything else, in flight first
  float bq[SKG_COLS];
  const float* bsrc = basis_t + (size_t)3 * v0 * n_coef + tid;
  const bool kon = tid < n_coef;
#pragma unroll
  for (int i = 0; i < SKG_COLS; ++i) bq[i] = (kon && i < ncol) ? bsrc[(size_t)i * n_coef] : 0.f;
  float w[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  float4 d = make_float4(0.f, 0.f, 0.f, 0.f), a = d;
  const bool von = tid < nv;
  if (von) {
    const int v = v0 + tid;
#pragma unroll
    for (int j = 0; j < 5; ++j) w[j] = lbs_weights[(size_t)v * 8 + j];
    d = reinterpret_cast<const float4*>(dverts)[v];
    a = reinterpret_cast<const float4*>(v_shaped)[v];
  }
  if (tid >= 64 && tid < 124) X[tid - 64] = joint_xf[tid - 64];
  // the forward half of the serial front, by EVERY workgroup in the shadow of the loads above (a few KB from L2, ~200 instructions
  // per thread): the workgroup that turns out to be the last then starts its chain backward at once instead of first regressing
  // the joints and evaluating five Rodrigues maps behind two more memory round trips
  __shared__ FrontLds F;
  flame_front_prepare(F, fa);              // (ends on a barrier: X and the staged loads are published as well)
  if (tid < SKG_V) {
    float out[3] = {0.f, 0.f, 0.f};
    const float dv[3] = {d.x, d.y, d.z};
    if (von) {
      reinterpret_cast<float4*>(dverts)[v0 + tid] = make_float4(0.f, 0.f, 0.f, 0.f);   // consumed
#pragma unroll
      for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int c = 0; c < 3; ++c) out[c] = fma_(X[j * 12 + r * 3 + c], w[j] * dv[r], out[c]);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) { s_dvs[3 * tid + c] = out[c]; s_dv[tid][c] = dv[c]; }
#pragma unroll
    for (int j = 0; j < 5; ++j) s_w[tid][j] = w[j];
    s_vs[tid][0] = a.x; s_vs[tid][1] = a.y; s_vs[tid][2] = a.z; s_vs[tid][3] = 1.f;
  }
  __syncthreads();
  if (kon) {
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < SKG_COLS; ++i) acc = fma_(bq[i], s_dvs[i], acc);
    atomicAdd(&dcoef[(blockIdx.x % SKG_G) * SKG_NT + tid], acc);
  }
  if (tid < 63) {
    float t = 0.f;
    if (tid < 60) {
      const int j = tid / 12, i = tid % 12, r = i < 9 ? i / 3 : i - 9, c = i < 9 ? i % 3 : 3;
      for (int v = 0; v < nv; ++v) t = fma_(s_w[v][j] * s_dv[v][r], s_vs[v][c], t);
    } else {
      for (int v = 0; v < nv; ++v) t += s_dv[v][tid - 60];
    }
    atomicAdd(&sums[(blockIdx.x % SKG_G) * 64 + tid], t);
  }
  // every wave's atomics have been performed (acknowledged by the memory side) before the ticket is drawn; float atomics are
  // agent-scope operations that bypass the caches, so no release fence -- a `buffer_wbl2` per thread -- is needed for them
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  uint32_t* ticket = reinterpret_cast<uint32_t*>(dcoef + SKG_G * SKG_NT);
  if (tid == 0) s_ticket = atomicAdd(ticket, 1u);
  __syncthreads();
  if (s_ticket != gridDim.x - 1) return;        // uniform over the block
  if (tid == 0) *ticket = 0u;
  // the totals are read where the adds were performed and consumed in the same operation: an atomic exchange with zero (the next
  // call's atomics start from zero) -- no cache between this workgroup and the other workgroups' adds
  __shared__ float s_tot[SKG_NT + 64];
  {
    float part[SKG_G];
#pragma unroll
    for (int g = 0; g < SKG_G; ++g) part[g] = tid < n_coef ? atomicExch(&dcoef[g * SKG_NT + tid], 0.f) : 0.f;
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < SKG_G; ++g) t += part[g];
    s_tot[tid] = t;
  }
  if (tid < 64) {
    float part[SKG_G];
#pragma unroll
    for (int g = 0; g < SKG_G; ++g) part[g] = tid < 63 ? atomicExch(&sums[g * 64 + tid], 0.f) : 0.f;
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < SKG_G; ++g) t += part[g];
    s_tot[SKG_NT + tid] = t;
  }
  __syncthreads();
  FrontArgs f2 = fa;
  f2.partial = s_tot + SKG_NT;
  flame_front_finish(F, f2, s_tot);
}

// axis-angle -> rotation matrix, the formula of flame_fitter.py:133-152: a = aa / (|aa| + 1e-8), R = I + sin K + (1 - cos) K^2
__device__ __forceinline__ void rodrigues_fwd(const float* aa, float* R, float* K, float& th) {
  th = sqrtf(aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2]);
  const float inv = 1.f / (th + 1e-8f);
  const float a0 = aa[0] * inv, a1 = aa[1] * inv, a2 = aa[2] * inv;
  const float Kl[9] = {0.f, -a2, a1, a2, 0.f, -a0, -a1, a0, 0.f};
  const float sn = sinf(th), cs = cosf(th);
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      float k2 = 0.f;
      for (int m = 0; m < 3; ++m) k2 = fma_(Kl[r * 3 + m], Kl[m * 3 + c], k2);
      R[r * 3 + c] = (r == c ? 1.f : 0.f) + sn * Kl[r * 3 + c] + (1.f - cs) * k2;
      K[r * 3 + c] = Kl[r * 3 + c];
    }
}

__global__ void rodrigues_kernel(const float* __restrict__ aa, int n, float* __restrict__ rotmats) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float R[9], K[9], th;
  rodrigues_fwd(aa + (size_t)i * 3, R, K, th);
  for (int k = 0; k < 9; ++k) rotmats[(size_t)i * 9 + k] = R[k];
}

// Gradient of the small FLAME front: (d joint_xf [5][12], d coef [K], d translation) -> (d expr [E], d pose [5][3]).
// One wave; lane 0 walks the 5-joint chain and the axis-angle maps backwards, all lanes finish d expr.
// Called by all threads of one block (barriers inside); dcoef was written by other blocks: read through volatile,
// once, into LDS (n_expr + 36 <= 256).
__device__ void flame_front_prepare(FrontLds& F, const FrontArgs& fa) {
  const float* j_static = fa.j_static; const float* j_expr = fa.j_expr; const float* expr = fa.expr; const float* pose = fa.pose;
  const int n_expr = fa.n_expr;
  const int lane = threadIdx.x;
  if (lane < 15) F.s_pose[lane] = pose[lane];
  for (int k = lane; k < 15 * n_expr; k += (int)blockDim.x) F.s_je[k] = j_expr[k];
  for (int k = lane; k < n_expr; k += (int)blockDim.x) F.s_e[k] = expr[k];
  __syncthreads();
  if (lane < 15) {
    float a = j_static[lane];
    const float* row = F.s_je + lane * n_expr;
    for (int k = 0; k < n_expr; ++k) a = fma_(row[k], F.s_e[k], a);
    F.sJ[lane] = a;
  }
  if (lane >= 64 && lane < 69) {       // the five joint rotations (and their generators), one lane each
    float Rj[9], Kj[9], t;
    rodrigues_fwd(F.s_pose + (lane - 64) * 3, Rj, Kj, t);
    for (int i = 0; i < 9; ++i) { F.sR[lane - 64][i] = Rj[i]; F.sK[lane - 64][i] = Kj[i]; }
    F.sth[lane - 64] = t;
  }
  __syncthreads();
}

__device__ void flame_front_finish(FrontLds& F, const FrontArgs& fa, const float* dcoef_) {
  const volatile float* dcoef = dcoef_;
  const int n_expr = fa.n_expr, n_rows = fa.n_rows;
  const float* partial = fa.partial;
  float* dexpr = fa.dexpr; float* dpose = fa.dpose; float* dtrans = fa.dtrans;
  float (&sJ)[15] = F.sJ; float (&sdJ)[15] = F.sdJ; float (&sums)[64] = F.sums; float (&part16)[16][64] = F.part16;
  float (&s_dcoef)[256] = F.s_dcoef; float (&s_pose)[15] = F.s_pose; float (&s_je)[15 * 128] = F.s_je;
  float (&sR)[5][9] = F.sR; float (&sK)[5][9] = F.sK; float (&sth)[5] = F.sth;
  const int lane = threadIdx.x;
  if (lane < n_expr + 36 && lane < 256) s_dcoef[lane] = dcoef[lane];
  if (fa.totals) {
    if (lane < 64) part16[0][lane] = const_cast<const volatile float*>(partial)[lane];
    for (int i = 64 + lane; i < 16 * 64; i += (int)blockDim.x) part16[0][i] = 0.f;
  } else {   // add up the per-wave rows of flame_skin_bwd: thread (w, q) of the 1024 sums value q over the rows r = w (mod 16), in row
      // order, eight loads in flight at a time (the 4-way form walked 21 rows per thread one memory round trip after the other)
    const int w = lane >> 6, q = lane & 63;
    float t = 0.f;
    for (int r0 = w; r0 < n_rows; r0 += 8 * 16) {
      float pv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int r = r0 + u * 16; pv[u] = r < n_rows ? partial[(size_t)r * 64 + q] : 0.f; }
#pragma unroll
      for (int u = 0; u < 8; ++u) t += pv[u];
    }
    part16[w][q] = t;
  }
  __syncthreads();
  if (lane < 64) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += part16[w][lane];
    sums[lane] = t;
    if (lane >= 60 && lane < 63) dtrans[lane - 60] = t;
  }
  __syncthreads();
  if (lane == 0) {
    float R[5][9], K[5][9], th[5], J[5][3];
    for (int j = 0; j < 5; ++j) {
      for (int i = 0; i < 9; ++i) { R[j][i] = sR[j][i]; K[j][i] = sK[j][i]; }
      th[j] = sth[j];
      for (int c = 0; c < 3; ++c) J[j][c] = sJ[j * 3 + c];
    }
    const int par[5] = {-1, 0, 1, 1, 1};
    float Rw[5][9], dRw[5][9], dtw[5][3], dJ[5][3], dR[5][9];
    for (int i = 0; i < 9; ++i) Rw[0][i] = R[0][i];
    for (int j = 1; j < 5; ++j)
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
          float a = 0.f;
          for (int m = 0; m < 3; ++m) a = fma_(Rw[par[j]][r * 3 + m], R[j][m * 3 + c], a);
          Rw[j][r * 3 + c] = a;
        }
    // X_j = [Rw_j | tw_j - Rw_j J_j]
    for (int j = 0; j < 5; ++j) {
      const float* dX = sums + j * 12;
      for (int r = 0; r < 3; ++r) {
        dtw[j][r] = dX[9 + r];
        for (int c = 0; c < 3; ++c) dRw[j][r * 3 + c] = dX[r * 3 + c] - dX[9 + r] * J[j][c];
      }
      for (int c = 0; c < 3; ++c) {
        float a = 0.f;
        for (int r = 0; r < 3; ++r) a = fma_(Rw[j][r * 3 + c], dX[9 + r], a);
        dJ[j][c] = -a;
      }
      for (int i = 0; i < 9; ++i) dR[j][i] = 0.f;
    }
    for (int j = 4; j >= 1; --j) {
      const int p = par[j];
      // tw_j = Rw_p (J_j - J_p) + tw_p
      for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) dRw[p][r * 3 + c] = fma_(dtw[j][r], J[j][c] - J[p][c], dRw[p][r * 3 + c]);
        dtw[p][r] += dtw[j][r];
      }
      for (int c = 0; c < 3; ++c) {
        float a = 0.f;
        for (int r = 0; r < 3; ++r) a = fma_(Rw[p][r * 3 + c], dtw[j][r], a);
        dJ[j][c] += a; dJ[p][c] -= a;
      }
      // Rw_j = Rw_p R_j
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
          float a = 0.f, b = 0.f;
          for (int m = 0; m < 3; ++m) {
            a = fma_(dRw[j][r * 3 + m], R[j][c * 3 + m], a);       // dRw_j R_j^T
            b = fma_(Rw[p][m * 3 + r], dRw[j][m * 3 + c], b);      // Rw_p^T dRw_j
          }
          dRw[p][r * 3 + c] += a;
          dR[j][r * 3 + c] += b;
        }
    }
    for (int i = 0; i < 9; ++i) dR[0][i] += dRw[0][i];
    for (int c = 0; c < 3; ++c) dJ[0][c] += dtw[0][c];
    // pose features (R_j - I), j = 1..4, are coefficients n_expr .. n_expr+35
    for (int j = 1; j < 5; ++j)
      for (int i = 0; i < 9; ++i) dR[j][i] += s_dcoef[n_expr + (j - 1) * 9 + i];
    for (int j = 0; j < 5; ++j)
      for (int i = 0; i < 9; ++i) sR[j][i] = dR[j][i];        // sR now carries dL/dR_j to the five lanes below
    for (int i = 0; i < 15; ++i) sdJ[i] = dJ[i / 3][i % 3];
  }
  __syncthreads();
  if (lane < 5) {          // axis-angle maps backwards, one joint per lane
    const int j = lane;
    {
      const float* G = sR[j];
      const float* Kj = sK[j];
      const float t = sth[j], sn = sinf(t), cs = cosf(t);
      float K2[9], dK[9];
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
          float a = 0.f;
          for (int m = 0; m < 3; ++m) a = fma_(Kj[r * 3 + m], Kj[m * 3 + c], a);
          K2[r * 3 + c] = a;
        }
      float dsn = 0.f, dom = 0.f;   // d/d sin, d/d (1 - cos)
      for (int i = 0; i < 9; ++i) { dsn = fma_(G[i], Kj[i], dsn); dom = fma_(G[i], K2[i], dom); }
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
          float a = 0.f;   // (G K^T + K^T G)[r][c]
          for (int m = 0; m < 3; ++m) a = fma_(G[r * 3 + m], Kj[c * 3 + m], fma_(Kj[m * 3 + r], G[m * 3 + c], a));
          dK[r * 3 + c] = fma_(1.f - cs, a, sn * G[r * 3 + c]);
        }
      float dth = cs * dsn + sn * dom;
      const float da[3] = {dK[7] - dK[5], dK[2] - dK[6], dK[3] - dK[1]};
      const float* aa = s_pose + j * 3;
      const float inv = 1.f / (t + 1e-8f);
      dth -= (da[0] * aa[0] + da[1] * aa[1] + da[2] * aa[2]) * inv * inv;
      for (int c = 0; c < 3; ++c) dpose[j * 3 + c] = da[c] * inv + (t > 0.f ? dth * aa[c] / t : 0.f);
    }
  }
  for (int e = lane; e < n_expr; e += (int)blockDim.x) {
    float a = s_dcoef[e];
    for (int i = 0; i < 15; ++i) a = fma_(s_je[i * n_expr + e], sdJ[i], a);
    dexpr[e] = a;
  }
}

// torch.optim.Adam semantics on a flat buffer (the FLAME parameter tensors)
__global__ void adam_flat_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int n,
                                 float lr_step, float b1, float b2, float eps, float inv_sqrt_bc2, float grad_scale) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float ge = g[i] * grad_scale;
  const float me = fma_(b1, m[i], (1.f - b1) * ge);
  const float ve = fma_(b2, v[i], (1.f - b2) * ge * ge);
  m[i] = me; v[i] = ve;
  p[i] = p[i] - lr_step * (me / fma_(sqrtf(ve), inv_sqrt_bc2, eps));
}

// Up to four flat tensors in one launch; the gradient is CONSUMED (zeroed after it is read), so dense gradient tensors of
// which one row is written per step need no clearing pass.
struct AdamSeg { float* p; float* g; float* m; float* v; int n; float lr_step; };
struct AdamSegs { AdamSeg s[4]; int n_seg; };
__global__ void adam_flat_multi_kernel(AdamSegs segs, float b1, float b2, float eps, float inv_sqrt_bc2, float grad_scale,
                                       const omfs_step_state* __restrict__ st) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  float lr_mul = 1.f;
  if (st) { lr_mul = st->flame_inv_bc1; inv_sqrt_bc2 = st->flame_inv_sqrt_bc2; }   // device-resident step (graph replay)
  for (int k = 0; k < segs.n_seg; ++k) {
    const AdamSeg& sg = segs.s[k];
    const int padded = (sg.n + 255) / 256 * 256;       // segments start on block boundaries
    if (i >= padded) { i -= padded; continue; }
    if (i < sg.n) {
      const float ge = sg.g[i] * grad_scale;
      sg.g[i] = 0.f;
      const float me = fma_(b1, sg.m[i], (1.f - b1) * ge);
      const float ve = fma_(b2, sg.v[i], (1.f - b2) * ge * ge);
      sg.m[i] = me; sg.v[i] = ve;
      sg.p[i] = sg.p[i] - (sg.lr_step * lr_mul) * (me / fma_(sqrtf(ve), inv_sqrt_bc2, eps));
    }
    return;                                             // the tail threads of a segment's last block own nothing
  }
}

}  // namespace omfs

using namespace omfs;

extern "C" int omfs_adam_flat_multi(int n_tensors, float* const* params, float* const* grads, float* const* m, float* const* v,
                                    const int* n, const float* lr, float beta1, float beta2, float eps, int step, float grad_scale,
                                    const omfs_step_state* state_dev, void* stream) {
  OMFS_REQUIRE(n_tensors >= 1 && n_tensors <= 4 && params && grads && m && v && n && lr && (state_dev || step >= 1), "args");
  const double bc1 = state_dev ? 1.0 : 1.0 - pow((double)beta1, step), bc2 = state_dev ? 1.0 : 1.0 - pow((double)beta2, step);
  AdamSegs segs;
  segs.n_seg = n_tensors;
  int blocks = 0;
  for (int k = 0; k < n_tensors; ++k) {
    OMFS_REQUIRE(params[k] && grads[k] && m[k] && v[k] && n[k] > 0, "tensor");
    segs.s[k] = AdamSeg{params[k], grads[k], m[k], v[k], n[k], (float)(lr[k] / bc1)};
    blocks += cdiv(n[k], 256);
  }
  hipLaunchKernelGGL(adam_flat_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, segs, beta1, beta2, eps,
                     (float)(1.0 / sqrt(bc2)), grad_scale, state_dev);
  OMFS_CHECK_HIP(hipGetLastError());
  return OMFS_OK;
}

extern "C" int omfs_flame_joints(const omfs_flame_rig* rig, const float* expr, const float* rotmats, int n_frames,
                                 float* joint_xf, float* coef, const int32_t* frame_index, void* stream) {
  OMFS_REQUIRE(rig && expr && rotmats && joint_xf && coef, "null pointer");
  OMFS_REQUIRE(n_frames > 0 && rig->n_expr > 0 && rig->n_expr <= 128 && rig->k_pad % 16 == 0 && rig->k_pad >= rig->n_expr + 36, "shape");
  int b_pad = cdiv(n_frames, 16) * 16;
  hipLaunchKernelGGL(flame_joints_kernel, dim3(b_pad), dim3(64), 0, (hipStream_t)stream, rig->j_static,
                     rig->j_expr, expr, const_cast<float*>(rotmats), (const float*)nullptr, n_frames, rig->n_expr, rig->k_pad, b_pad,
                     joint_xf, coef, frame_index);
  OMFS_CHECK_HIP(hipGetLastError());
  return OMFS_OK;
}

extern "C" int omfs_flame_joints_pose(const omfs_flame_rig* rig, const float* expr, const float* pose, float* rotmats, int n_frames,
                                      float* joint_xf, float* coef, const int32_t* frame_index, void* stream) {
  OMFS_REQUIRE(rig && expr && pose && rotmats && joint_xf && coef, "null pointer");
  OMFS_REQUIRE(n_frames > 0 && rig->n_expr > 0 && rig->n_expr <= 128 && rig->k_pad % 16 == 0 && rig->k_pad >= rig->n_expr + 36, "shape");
  int b_pad = cdiv(n_frames, 16) * 16;
  hipLaunchKernelGGL(flame_joints_kernel, dim3(b_pad), dim3(64), 0, (hipStream_t)stream, rig->j_static,
                     rig->j_expr, expr, rotmats, pose, n_frames, rig->n_expr, rig->k_pad, b_pad, joint_xf, coef, frame_index);
  OMFS_CHECK_HIP(hipGetLastError());
  return OMFS_OK;
}

extern "C" int omfs_flame_lbs(const omfs_flame_rig* rig, const float* coef, const float* joint_xf,
                              const float* translation, const float* dynamic_offset, int n_frames, float* verts,
                              float* v_shaped_out, const int32_t* frame_index, void* stream) {
  OMFS_REQUIRE(rig && coef && joint_xf && translation && verts, "null pointer");
  OMFS_REQUIRE(n_frames > 0 && rig->v_pad % 16 == 0 && rig->v_pad >= rig->n_verts && rig->k_pad % 16 == 0, "shape");
  int b_pad = cdiv(n_frames, 16) * 16;
  hipLaunchKernelGGL(flame_lbs_kernel, dim3(rig->v_pad / 16, b_pad / 16), dim3(64), 0, (hipStream_t)stream,
                     rig->basis_tiled, rig->v_static, rig->lbs_weights, coef, joint_xf, translation, dynamic_offset,
                     rig->n_verts, rig->v_pad, rig->k_pad, n_frames, b_pad, verts, v_shaped_out, frame_index);
  OMFS_CHECK_HIP(hipGetLastError());
  return OMFS_OK;
}

extern "C" int omfs_flame_pose_lbs(const omfs_flame_rig* rig, const float* expr, float* rotmats, const float* pose,
                                   const float* translation, const float* dynamic_offset, float* joint_xf, float* coef,
                                   float* verts, float* v_shaped_out, const int32_t* frame_index, void* stream) {
  OMFS_REQUIRE(rig && expr && rotmats && translation && joint_xf && coef && verts, "null pointer");
  OMFS_REQUIRE(rig->n_expr > 0 && rig->n_expr <= 128 && rig->k_pad % 16 == 0 && rig->k_pad >= rig->n_expr + 36 && rig->k_pad <= 192 &&
                   rig->v_pad % 16 == 0 && rig->v_pad >= rig->n_verts, "shape");
  hipLaunchKernelGGL(flame_pose_lbs_kernel, dim3(rig->v_pad / 16), dim3(64), 0, (hipStream_t)stream, rig->basis_tiled, rig->v_static,
                     rig->lbs_weights, rig->j_static, rig->j_expr, expr, rotmats, pose, translation, dynamic_offset, rig->n_verts,
                     rig->v_pad, rig->k_pad, rig->n_expr, verts, v_shaped_out, joint_xf, coef, frame_index);
  OMFS_CHECK_HIP(hipGetLastError());
  return OMFS_OK;
}

extern "C" int omfs_face_frames(const float* verts, int v_pad, const int32_t* faces, int n_faces, int n_frames,
                                float* face_xf, void* stream) {
  OMFS_REQUIRE(verts && faces && face_xf, "null pointer");
  OMFS_REQUIRE(n_faces > 0 && n_frames > 0 && v_pad > 0, "shape");
  hipLaunchKernelGGL(face_frames_kernel, dim3(cdiv(n_faces, 256), n_frames), dim3(256), 0, (hipStream_t)stream, verts,
                     v_pad, faces, n_faces, n_frames, face_xf);
  OMFS_CHECK_HIP(hipGetLastError());
  return OMFS_OK;
}

extern "C" int omfs_face_frames_bwd(const float* verts, int v_pad, const int32_t* faces, int n_faces, const float* dface,
                                    const int32_t* face_start, const int32_t* face_gauss, float* dverts, void* stream) {
  OMFS_REQUIRE(verts && faces && dface && face_start && face_gauss && dverts, "null pointer");
  OMFS_REQUIRE(n_faces > 0 && v_pad > 0, "shape");
  hipLaunchKernelGGL(face_frames_bwd_kernel<false>, dim3(cdiv(n_faces * 16, 256)), dim3(256), 0, (hipStream_t)stream, verts, faces,
                     n_faces, dface, face_start, face_gauss, dverts, (long long*)nullptr);
  OMFS_CHECK_HIP(hipGetLastError());
  return OMFS_OK;
}

extern "C" int omfs_face_frames_bwd_fx(const float* verts, int v_pad, const int32_t* faces, int n_faces, const float* dface,
                                       const int32_t* face_start, const int32_t* face_gauss, float* dverts, long long* dverts_fx, void* stream) {
  OMFS_REQUIRE(verts && faces && dface && face_start && face_gauss && dverts && dverts_fx, "null pointer");
  OMFS_REQUIRE(n_faces > 0 && v_pad > 0, "shape");
  hipLaunchKernelGGL(face_frames_bwd_kernel<true>, dim3(cdiv(n_faces * 16, 256)), dim3(256), 0, (hipStream_t)stream, verts, faces,
                     n_faces, dface, face_start, face_gauss, dverts, dverts_fx);
  OMFS_CHECK_HIP(hipGetLastError());
  hipLaunchKernelGGL(dverts_from_fixed_kernel, dim3(cdiv(v_pad * 4, 256)), dim3(256), 0, (hipStream_t)stream, dverts_fx, dverts, v_pad * 4);
  OMFS_CHECK_HIP(hipGetLastError());
  return OMFS_OK;
}

extern "C" int omfs_flame_skin_rows(const omfs_flame_rig* rig) { return rig ? cdiv(rig->n_verts, 256) * 4 : 0; }

extern "C" int omfs_flame_skin_bwd(const omfs_flame_rig* rig, const float* v_shaped, const float* joint_xf, float* dverts,
                                   float* dv_shaped, float* sums, void* stream) {
  OMFS_REQUIRE(rig && v_shaped && joint_xf && dverts && dv_shaped && sums, "null pointer");
  OMFS_REQUIRE(rig->n_verts > 0 && rig->lbs_weights, "rig");
  hipLaunchKernelGGL(flame_skin_bwd_kernel, dim3(cdiv(rig->n_verts, 256)), dim3(256), 0, (hipStream_t)stream,
                     rig->lbs_weights, v_shaped, joint_xf, dverts, rig->n_verts, dv_shaped, sums);
  OMFS_CHECK_HIP(hipGetLastError());
  return OMFS_OK;
}

extern "C" int omfs_flame_rodrigues(const float* axis_angle, int n, float* rotmats, void* stream) {
  OMFS_REQUIRE(axis_angle && rotmats && n > 0, "args");
  hipLaunchKernelGGL(rodrigues_kernel, dim3(cdiv(n, 64)), dim3(64), 0, (hipStream_t)stream, axis_angle, n, rotmats);
  OMFS_CHECK_HIP(hipGetLastError());
  return OMFS_OK;
}

extern "C" int omfs_flame_param_bwd(const omfs_flame_rig* rig, const float* basis_dense, int n_coef, const float* dv_shaped,
                                    const float* expr, const float* pose, const float* sums, float* dcoef, float* dexpr,
                                    float* dpose, float* dtrans, void* stream) {
  OMFS_REQUIRE(rig && basis_dense && dv_shaped && expr && pose && sums && dcoef && dexpr && dpose && dtrans, "null pointer");
  OMFS_REQUIRE(n_coef == rig->n_expr + 36 && n_coef <= 256 && rig->n_expr <= 128 && rig->j_static && rig->j_expr, "shape");
  hipStream_t s = (hipStream_t)stream;
  FrontArgs fa{rig->j_static, rig->j_expr, expr, pose, rig->n_expr, sums, cdiv(rig->n_verts, 256) * 4, dexpr, dpose, dtrans, 0};
  hipLaunchKernelGGL(basis_t_gemv_kernel, dim3(n_coef), dim3(GEMV_NT), 0, s, basis_dense, dv_shaped, 3 * rig->n_verts, dcoef, fa);
  OMFS_CHECK_HIP(hipGetLastError());
  return OMFS_OK;
}

extern "C" int omfs_flame_skin_param_scratch_floats(int which) { return which == 0 ? SKG_G * SKG_NT + 4 : SKG_G * 64; }

extern "C" int omfs_flame_skin_param_bwd(const omfs_flame_rig* rig, const float* basis_t, int n_coef, const float* v_shaped,
                                         const float* joint_xf, float* dverts, const float* expr, const float* pose,
                                         float* dcoef, float* sums, float* dexpr, float* dpose, float* dtrans, void* stream) {
  OMFS_REQUIRE(rig && basis_t && v_shaped && joint_xf && dverts && expr && pose && dcoef && sums && dexpr && dpose && dtrans, "null pointer");
  OMFS_REQUIRE(n_coef == rig->n_expr + 36 && n_coef <= SKG_NT && rig->n_expr <= 128 && rig->n_verts > 0 && rig->lbs_weights &&
                   rig->j_static && rig->j_expr, "shape");
  FrontArgs fa{rig->j_static, rig->j_expr, expr, pose, rig->n_expr, sums, 1, dexpr, dpose, dtrans, 1};
  hipLaunchKernelGGL(flame_skin_gemv_kernel, dim3(cdiv(rig->n_verts, SKG_V)), dim3(SKG_NT), 0, (hipStream_t)stream, rig->lbs_weights,
                     v_shaped, joint_xf, dverts, rig->n_verts, basis_t, n_coef, dcoef, sums, fa);
  OMFS_CHECK_HIP(hipGetLastError());
  return OMFS_OK;
}

extern "C" int omfs_adam_flat(float* params, const float* grads, float* m, float* v, int n, float lr, float beta1, float beta2,
                              float eps, int step, float grad_scale, void* stream) {
  OMFS_REQUIRE(params && grads && m && v && n > 0 && step >= 1, "args");
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  hipLaunchKernelGGL(adam_flat_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, params, grads, m, v, n,
                     (float)(lr / bc1), beta1, beta2, eps, (float)(1.0 / sqrt(bc2)), grad_scale);
  OMFS_CHECK_HIP(hipGetLastError());
  return OMFS_OK;
}

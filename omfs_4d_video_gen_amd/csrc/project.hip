// Per-Gaussian forward: triangle-bound deformation, EWA projection, SH colour, tile rectangle,
// (hit counts moved to binning.hip).  One lane per Gaussian, planar SoA loads (256 contiguous bytes per wave
// instruction and plane), three coalesced float4 stores.  HBM-streaming kernel:
// algorithmic bytes per Gaussian = 59*4 (params) + 4 (binding) + 48 (records) [+ 64 face record, L2].
//
// Arithmetic of the geometric part follows DESIGN.md "Frozen arithmetic" operation by operation
// (explicit fma placement, IEEE div/sqrt, exp_exact), so that radii and tile rectangles are
// bit-identical to oracle/splat_oracle.c.  Built with -ffp-contract=off.
//
// Spec: SURVEY.md Appendix A items 3-4 (absent upstream rasteriser; call site
// 02_Visual_Engine/render_surgery.py:289-315 / train_ghost.py:227-271).
#include "common.hpp"

namespace omfs {

constexpr float SH_C0 = 0.28209479177387814f;
constexpr float SH_C1 = 0.4886025119029199f;

struct ProjCam {
  float view[12];
  float cam_pos[3];
  float fx, fy, cx, cy, limx, limy;
  int width, height, gx, gy, sh_degree;
};

__device__ __forceinline__ float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

__global__ __launch_bounds__(256) void project_fwd_kernel(int n, int n_pad, const float* __restrict__ params,
                                                          const int32_t* __restrict__ binding,
                                                          const float* __restrict__ face_xf, ProjCam cam,
                                                          float4* __restrict__ g0, float4* __restrict__ g1,
                                                          float4* __restrict__ g2, uint32_t* __restrict__ n_visible,
                                                          uint32_t* __restrict__ status) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (n_visible && i == 0) n_visible[0] = 0u;      // accumulated by omfs_bin_count, which runs after this kernel
  // new projected records: tile-test ballots an earlier omfs_bin_count left for omfs_bin_scatter (replay stamp, status[1]) were
  // taken from OTHER records -- same parameter pointer and camera, e.g. the previous iteration's -- and must not be replayed
  if (status && i == 0) status[1] = 0u;
  if (i >= n) return;
  auto P = [&](int plane) { return params[(size_t)plane * n_pad + i]; };

  // ---- parent triangle frame (64-byte record, L2-resident table)
  const float4* fr = reinterpret_cast<const float4*>(face_xf) + (size_t)binding[i] * 4;
  const float4 f0 = fr[0], f1 = fr[1], f2 = fr[2], f3 = fr[3];
  const float R00 = f0.x, R01 = f0.y, R02 = f0.z, R10 = f0.w, R11 = f1.x, R12 = f1.y, R20 = f1.z, R21 = f1.w, R22 = f2.x;
  const float cfx = f2.y, cfy = f2.z, cfz = f2.w, sf = f3.x;

  // ---- world mean
  const float lx = P(OMFS_P_XYZ + 0), ly = P(OMFS_P_XYZ + 1), lz = P(OMFS_P_XYZ + 2);
  const float mx = fma_(dot3_(R00, R01, R02, lx, ly, lz), sf, cfx);
  const float my = fma_(dot3_(R10, R11, R12, lx, ly, lz), sf, cfy);
  const float mz = fma_(dot3_(R20, R21, R22, lx, ly, lz), sf, cfz);

  // ---- view space
  const float* W = cam.view;
  const float tx = dot3_(W[0], W[1], W[2], mx, my, mz) + W[3];
  const float ty = dot3_(W[4], W[5], W[6], mx, my, mz) + W[7];
  const float tz = dot3_(W[8], W[9], W[10], mx, my, mz) + W[11];

  float4 o0 = make_float4(0.f, 0.f, 0.f, 0.f), o1 = o0, o2 = o0;
  bool visible = tz > 0.2f;
  if (visible) {
    // ---- world rotation / covariance
    float qw = P(OMFS_P_ROT + 0), qx = P(OMFS_P_ROT + 1), qy = P(OMFS_P_ROT + 2), qz = P(OMFS_P_ROT + 3);
    const float qn = sqrtf(fma_(qz, qz, fma_(qy, qy, fma_(qx, qx, qw * qw))));
    qw = qw / qn; qx = qx / qn; qy = qy / qn; qz = qz / qn;
    const float Q00 = 1.f - 2.f * fma_(qy, qy, qz * qz), Q01 = 2.f * fma_(qx, qy, -(qw * qz)), Q02 = 2.f * fma_(qx, qz, qw * qy);
    const float Q10 = 2.f * fma_(qx, qy, qw * qz), Q11 = 1.f - 2.f * fma_(qx, qx, qz * qz), Q12 = 2.f * fma_(qy, qz, -(qw * qx));
    const float Q20 = 2.f * fma_(qx, qz, -(qw * qy)), Q21 = 2.f * fma_(qy, qz, qw * qx), Q22 = 1.f - 2.f * fma_(qx, qx, qy * qy);
    const float s0 = exp_exact(P(OMFS_P_SCALE + 0)) * sf, s1 = exp_exact(P(OMFS_P_SCALE + 1)) * sf,
                s2 = exp_exact(P(OMFS_P_SCALE + 2)) * sf;
    // M = (R_f Q) diag(s)
    const float M00 = dot3_(R00, R01, R02, Q00, Q10, Q20) * s0, M01 = dot3_(R00, R01, R02, Q01, Q11, Q21) * s1, M02 = dot3_(R00, R01, R02, Q02, Q12, Q22) * s2;
    const float M10 = dot3_(R10, R11, R12, Q00, Q10, Q20) * s0, M11 = dot3_(R10, R11, R12, Q01, Q11, Q21) * s1, M12 = dot3_(R10, R11, R12, Q02, Q12, Q22) * s2;
    const float M20 = dot3_(R20, R21, R22, Q00, Q10, Q20) * s0, M21 = dot3_(R20, R21, R22, Q01, Q11, Q21) * s1, M22 = dot3_(R20, R21, R22, Q02, Q12, Q22) * s2;
    const float S00 = dot3_(M00, M01, M02, M00, M01, M02), S01 = dot3_(M00, M01, M02, M10, M11, M12), S02 = dot3_(M00, M01, M02, M20, M21, M22);
    const float S11 = dot3_(M10, M11, M12, M10, M11, M12), S12 = dot3_(M10, M11, M12, M20, M21, M22), S22 = dot3_(M20, M21, M22, M20, M21, M22);

    // ---- projection
    const float xz = tx / tz, yz = ty / tz;
    const float px = fma_(cam.fx, xz, cam.cx), py = fma_(cam.fy, yz, cam.cy);
    const float txc = clampf(xz, -cam.limx, cam.limx) * tz, tyc = clampf(yz, -cam.limy, cam.limy) * tz;
    const float tz2 = tz * tz;
    const float J00 = cam.fx / tz, J02 = -(cam.fx * txc) / tz2, J11 = cam.fy / tz, J12 = -(cam.fy * tyc) / tz2;
    const float T00 = fma_(J02, W[8], J00 * W[0]), T01 = fma_(J02, W[9], J00 * W[1]), T02 = fma_(J02, W[10], J00 * W[2]);
    const float T10 = fma_(J12, W[8], J11 * W[4]), T11 = fma_(J12, W[9], J11 * W[5]), T12 = fma_(J12, W[10], J11 * W[6]);
    const float u0 = dot3_(S00, S01, S02, T00, T01, T02), u1 = dot3_(S01, S11, S12, T00, T01, T02), u2 = dot3_(S02, S12, S22, T00, T01, T02);
    const float w0 = dot3_(S00, S01, S02, T10, T11, T12), w1 = dot3_(S01, S11, S12, T10, T11, T12), w2 = dot3_(S02, S12, S22, T10, T11, T12);
    const float a = dot3_(T00, T01, T02, u0, u1, u2) + 0.3f;
    const float b = dot3_(T10, T11, T12, u0, u1, u2);
    const float c = dot3_(T10, T11, T12, w0, w1, w2) + 0.3f;
    const float det = fma_(a, c, -(b * b));
    visible = det != 0.f;
    if (visible) {
      const float mid = 0.5f * (a + c);
      const float lam = mid + sqrtf(fmaxf(0.1f, fma_(mid, mid, -det)));
      const float radius = fminf(ceilf(3.f * sqrtf(lam)), 1048575.f);
      const int x0 = min(cam.gx, max(0, (int)clampf((px - radius) / 16.f, -1.f, 4096.f)));
      const int y0 = min(cam.gy, max(0, (int)clampf((py - radius) / 16.f, -1.f, 4096.f)));
      const int x1 = min(cam.gx, max(0, (int)clampf(((px + radius) + 15.f) / 16.f, -1.f, 4096.f)));
      const int y1 = min(cam.gy, max(0, (int)clampf(((py + radius) + 15.f) / 16.f, -1.f, 4096.f)));
      visible = (x1 - x0) * (y1 - y0) > 0;
      if (visible) {
        // ---- colour (tolerance-level arithmetic from here on)
        float dx = mx - cam.cam_pos[0], dy = my - cam.cam_pos[1], dz = mz - cam.cam_pos[2];
        const float dl = sqrtf(fmaxf(dot3_(dx, dy, dz, dx, dy, dz), 1e-20f));
        dx = dx / dl; dy = dy / dl; dz = dz / dl;
        float rgb[3];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
          auto S = [&](int k) { return P(OMFS_P_SH + 3 * k + ch); };
          float r = SH_C0 * S(0);
          if (cam.sh_degree > 0) {
            r = r - SH_C1 * dy * S(1) + SH_C1 * dz * S(2) - SH_C1 * dx * S(3);
            if (cam.sh_degree > 1) {
              const float xx = dx * dx, yy = dy * dy, zz = dz * dz, xy = dx * dy, yz2 = dy * dz, xz2 = dx * dz;
              r = r + 1.0925484305920792f * xy * S(4) + -1.0925484305920792f * yz2 * S(5) +
                  0.31539156525252005f * (2.f * zz - xx - yy) * S(6) + -1.0925484305920792f * xz2 * S(7) +
                  0.5462742152960396f * (xx - yy) * S(8);
              if (cam.sh_degree > 2) {
                r = r + -0.5900435899266435f * dy * (3.f * xx - yy) * S(9) + 2.890611442640554f * xy * dz * S(10) +
                    -0.4570457994644658f * dy * (4.f * zz - xx - yy) * S(11) +
                    0.3731763325901154f * dz * (2.f * zz - 3.f * xx - 3.f * yy) * S(12) +
                    -0.4570457994644658f * dx * (4.f * zz - xx - yy) * S(13) + 1.445305721320277f * dz * (xx - yy) * S(14) +
                    -0.5900435899266435f * dx * (xx - 3.f * yy) * S(15);
              }
            }
          }
          rgb[ch] = r + 0.5f;
        }
        uint32_t clampbits = (rgb[0] < 0.f ? 1u : 0u) | (rgb[1] < 0.f ? 2u : 0u) | (rgb[2] < 0.f ? 4u : 0u);
        const float opac = 1.f / (1.f + exp_exact(-P(OMFS_P_OPACITY)));  // frozen: feeds the tile test
        o0 = make_float4(px, py, c / det, -b / det);
        o1 = make_float4(a / det, opac, fmaxf(rgb[0], 0.f), fmaxf(rgb[1], 0.f));
        o2 = make_float4(fmaxf(rgb[2], 0.f), tz, __uint_as_float((uint32_t)radius | (clampbits << 28)),
                         __uint_as_float((uint32_t)x0 | ((uint32_t)y0 << 8) | ((uint32_t)x1 << 16) | ((uint32_t)y1 << 24)));
      }
    }
  }
  g0[RI(i)] = o0;
  g1[RI(i)] = o1;
  g2[RI(i)] = o2;
}

__global__ __launch_bounds__(1024) void count_visible_kernel(const float4* __restrict__ g2, int n, uint32_t* __restrict__ out) {
  __shared__ uint32_t cnt;
  if (threadIdx.x == 0) cnt = 0;
  __syncthreads();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t vis = (i < n) && ((__float_as_uint(g2[RI(i)].z) & 0xFFFFFu) != 0u);
  const unsigned long long m = __ballot(vis);
  if (lane_id() == 0 && m) atomicAdd(&cnt, (uint32_t)__popcll(m));
  __syncthreads();
  if (threadIdx.x == 0 && cnt) atomicAdd(out, cnt);   // one global atomic per 1024 Gaussians
}

ProjCam make_projcam(const omfs_camera* c) {
  ProjCam p;
  for (int i = 0; i < 12; ++i) p.view[i] = c->view[i];
  for (int i = 0; i < 3; ++i) p.cam_pos[i] = c->cam_pos[i];
  p.fx = c->fx; p.fy = c->fy; p.cx = c->cx; p.cy = c->cy; p.limx = c->limx; p.limy = c->limy;
  p.width = c->width; p.height = c->height;
  p.gx = cdiv(c->width, OMFS_TILE); p.gy = cdiv(c->height, OMFS_TILE);
  p.sh_degree = c->sh_degree;
  return p;
}

}  // namespace omfs

using namespace omfs;

extern "C" int omfs_project_fwd(const omfs_gaussians* g, const float* face_xf, const omfs_camera* cam,
                                const omfs_raster_buffers* rb, void* stream) {
  OMFS_REQUIRE(g && face_xf && cam && rb, "null pointer");
  OMFS_REQUIRE(g->n > 0 && g->n_pad >= g->n && g->params && g->binding, "gaussians");
  OMFS_REQUIRE(cam->width > 0 && cam->height > 0 && cam->width <= 4080 && cam->height <= 4080, "image size (tile coords are 8 bit)");
  OMFS_REQUIRE(cam->sh_degree >= 0 && cam->sh_degree <= 3, "sh_degree");
  OMFS_REQUIRE(rb->g0 && rb->g1 && rb->g2, "raster buffers");
  OMFS_REQUIRE(OMFS_REC_STRIDE == 1 || (rb->g1 == rb->g0 + 4 && rb->g2 == rb->g0 + 8),
               "this library reads ONE 64-byte record per Gaussian: g1 = g0 + 4 floats, g2 = g0 + 8 floats (omfs_record_stride() == 4)");
  ProjCam pc = make_projcam(cam);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(project_fwd_kernel, dim3(cdiv(g->n, 256)), dim3(256), 0, s, g->n, g->n_pad, g->params, g->binding,
                     face_xf, pc, (float4*)rb->g0, (float4*)rb->g1, (float4*)rb->g2, rb->n_visible, rb->status);
  OMFS_CHECK_HIP(hipGetLastError());
  return OMFS_OK;
}

extern "C" int omfs_count_visible(const omfs_raster_buffers* rb, int n, uint32_t* count_out, void* stream) {
  OMFS_REQUIRE(rb && rb->g2 && count_out && n > 0, "args");
  hipStream_t s = (hipStream_t)stream;
  OMFS_CHECK_HIP(hipMemsetAsync(count_out, 0, sizeof(uint32_t), s));
  hipLaunchKernelGGL(count_visible_kernel, dim3(cdiv(n, 1024)), dim3(1024), 0, s, (const float4*)rb->g2, n, count_out);
  OMFS_CHECK_HIP(hipGetLastError());
  return OMFS_OK;
}

extern "C" int omfs_record_stride(void) { return OMFS_REC_STRIDE; }

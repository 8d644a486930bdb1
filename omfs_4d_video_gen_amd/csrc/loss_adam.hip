// Photometric loss (L1 + D-SSIM) forward+backward in image space, and the fused per-Gaussian Adam.
//
// Loss spec: SURVEY.md Appendix A item 8: (1-l) L1 + l (1 - SSIM), SSIM with an 11x11 Gaussian
// window (sigma 1.5), zero padding, per channel, C1 = 0.01^2, C2 = 0.03^2, mean over 3HW.
// Two kernels, each a separable 11-tap convolution through LDS on 16x16 tiles with a 5-pixel halo:
//   ssim_fwd : moments -> SSIM value + the three partial-derivative maps (d/dmu1, d/dE[xx], d/dE[xy])
//   ssim_bwd : convolve the maps back, add the L1 sign term -> dL/dimage
// Adam spec: SURVEY.md Appendix A item 9, torch.optim.Adam semantics; one pass over [59][n_pad],
// 16 bytes per lane: reads p,g,m,v, writes p,m,v = 7 x 4 B per element (HBM-streaming).
#include "common.hpp"

namespace omfs {

constexpr int LT = 16;        // output tile
constexpr int HALO = 5;
constexpr int LW = LT + 2 * HALO;  // 26

struct GaussW { float g[11]; };  // normalised 11-tap window, passed by value (scalar registers)

__global__ __launch_bounds__(256) void ssim_fwd_kernel(const float* __restrict__ img, const float* __restrict__ gt,
                                                       int width, int height, float w_l1, float w_ssim, GaussW gw,
                                                       float* __restrict__ map_mu1, float* __restrict__ map_xx,
                                                       float* __restrict__ map_xy, float* __restrict__ partials) {
  __shared__ float sx[LW][LW + 1], sy[LW][LW + 1];
  __shared__ float h[5][LW][LT + 1];
  __shared__ float wsum[4];
  const int ch = blockIdx.z, tid = threadIdx.x;
  const int ox = blockIdx.x * LT, oy = blockIdx.y * LT;
  const size_t plane = (size_t)width * height;
  const float* ip = img + ch * plane;
  const float* gp = gt + ch * plane;
  for (int k = tid; k < LW * LW; k += 256) {
    const int r = k / LW, c = k % LW;
    const int y = oy + r - HALO, x = ox + c - HALO;
    const bool in = x >= 0 && x < width && y >= 0 && y < height;
    sx[r][c] = in ? ip[(size_t)y * width + x] : 0.f;
    sy[r][c] = in ? gp[(size_t)y * width + x] : 0.f;
  }
  __syncthreads();
  for (int k = tid; k < LW * LT; k += 256) {
    const int r = k / LT, c = k % LT;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f;
#pragma unroll
    for (int t = 0; t < 11; ++t) {
      const float g = gw.g[t], x = sx[r][c + t], y = sy[r][c + t];
      a0 = fma_(g, x, a0); a1 = fma_(g, y, a1); a2 = fma_(g, x * x, a2); a3 = fma_(g, y * y, a3); a4 = fma_(g, x * y, a4);
    }
    h[0][r][c] = a0; h[1][r][c] = a1; h[2][r][c] = a2; h[3][r][c] = a3; h[4][r][c] = a4;
  }
  __syncthreads();
  const int lx = tid & 15, ly = tid >> 4;
  const int x = ox + lx, y = oy + ly;
  float contrib = 0.f;
  if (x < width && y < height) {
    float mu1 = 0.f, mu2 = 0.f, exx = 0.f, eyy = 0.f, exy = 0.f;
#pragma unroll
    for (int t = 0; t < 11; ++t) {
      const float g = gw.g[t];
      mu1 = fma_(g, h[0][ly + t][lx], mu1); mu2 = fma_(g, h[1][ly + t][lx], mu2);
      exx = fma_(g, h[2][ly + t][lx], exx); eyy = fma_(g, h[3][ly + t][lx], eyy); exy = fma_(g, h[4][ly + t][lx], exy);
    }
    const float C1 = 0.0001f, C2 = 0.0009f;
    const float s11 = exx - mu1 * mu1, s22 = eyy - mu2 * mu2, s12 = exy - mu1 * mu2;
    const float A1 = 2.f * mu1 * mu2 + C1, A2 = 2.f * s12 + C2, B1 = mu1 * mu1 + mu2 * mu2 + C1, B2 = s11 + s22 + C2;
    const float iB = 1.f / (B1 * B2);
    const float ssim = A1 * A2 * iB;
    const size_t o = ch * plane + (size_t)y * width + x;
    map_mu1[o] = (2.f * mu2 * (A2 - A1)) * iB - 2.f * mu1 * ssim / B1 + 2.f * mu1 * ssim / B2;
    map_xx[o] = -ssim / B2;
    map_xy[o] = 2.f * A1 * iB;
    const float d = sx[ly + HALO][lx + HALO] - sy[ly + HALO][lx + HALO];
    contrib = w_l1 * fabsf(d) - w_ssim * ssim;
  }
  contrib = wave_sum_all(contrib);
  if ((tid & 63) == 0) wsum[tid >> 6] = contrib;
  __syncthreads();
  // one partial per block (24k same-address atomics would serialise at ~11 ns each)
  if (tid == 0) partials[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

__global__ __launch_bounds__(256) void ssim_bwd_kernel(const float* __restrict__ img, const float* __restrict__ gt,
                                                       int width, int height, float w_l1, float w_ssim, GaussW gw,
                                                       const float* __restrict__ map_mu1, const float* __restrict__ map_xx,
                                                       const float* __restrict__ map_xy, float* __restrict__ dimage) {
  __shared__ float s[3][LW][LW + 1];
  __shared__ float h[3][LW][LT + 1];
  const int ch = blockIdx.z, tid = threadIdx.x;
  const int ox = blockIdx.x * LT, oy = blockIdx.y * LT;
  const size_t plane = (size_t)width * height;
  for (int k = tid; k < LW * LW; k += 256) {
    const int r = k / LW, c = k % LW;
    const int y = oy + r - HALO, x = ox + c - HALO;
    const bool in = x >= 0 && x < width && y >= 0 && y < height;
    const size_t o = ch * plane + (size_t)y * width + x;
    s[0][r][c] = in ? map_mu1[o] : 0.f;
    s[1][r][c] = in ? map_xx[o] : 0.f;
    s[2][r][c] = in ? map_xy[o] : 0.f;
  }
  __syncthreads();
  for (int k = tid; k < LW * LT; k += 256) {
    const int r = k / LT, c = k % LT;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int t = 0; t < 11; ++t) {
      const float g = gw.g[t];
      a0 = fma_(g, s[0][r][c + t], a0); a1 = fma_(g, s[1][r][c + t], a1); a2 = fma_(g, s[2][r][c + t], a2);
    }
    h[0][r][c] = a0; h[1][r][c] = a1; h[2][r][c] = a2;
  }
  __syncthreads();
  const int lx = tid & 15, ly = tid >> 4;
  const int x = ox + lx, y = oy + ly;
  if (x < width && y < height) {
    float c0 = 0.f, c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int t = 0; t < 11; ++t) {
      const float g = gw.g[t];
      c0 = fma_(g, h[0][ly + t][lx], c0); c1 = fma_(g, h[1][ly + t][lx], c1); c2 = fma_(g, h[2][ly + t][lx], c2);
    }
    const size_t o = ch * plane + (size_t)y * width + x;
    const float xv = img[o], yv = gt[o];
    const float d = xv - yv;
    const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    dimage[o] = w_l1 * sgn - w_ssim * (c0 + 2.f * xv * c1 + yv * c2);
  }
}

// one block: loss_out += constant + sum(partials)   (fixed order: bitwise reproducible)
__global__ __launch_bounds__(1024) void loss_reduce_kernel(const float* __restrict__ partials, int n, float constant,
                                                          float* __restrict__ loss_out) {
  __shared__ float ws[16];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += 1024) acc += partials[i];
  acc = wave_sum_all(acc);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < 16; ++w) t += ws[w];
    loss_out[0] += constant + t;
  }
}

struct AdamK {
  float lr_step[OMFS_NPLANES];  // lr / (1 - b1^t)
  float b1, b2, eps, inv_sqrt_bc2, grad_scale;
};

// grid = (n_pad/1024 [float4 x 256], 59)
__global__ __launch_bounds__(256) void adam_kernel(float4* __restrict__ p, const float4* __restrict__ g,
                                                   float4* __restrict__ m, float4* __restrict__ v, int n4_per_plane,
                                                   AdamK k) {
  const int plane = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4_per_plane) return;
  const size_t o = (size_t)plane * n4_per_plane + i;
  const float lr = k.lr_step[plane];
  float4 pp = p[o], gg = g[o], mm = m[o], vv = v[o];
  auto upd = [&](float& pe, float ge, float& me, float& ve) {
    ge *= k.grad_scale;
    me = fma_(k.b1, me, (1.f - k.b1) * ge);
    ve = fma_(k.b2, ve, (1.f - k.b2) * ge * ge);
    const float denom = fma_(sqrtf(ve), k.inv_sqrt_bc2, k.eps);
    pe = pe - lr * (me / denom);
  };
  upd(pp.x, gg.x, mm.x, vv.x);
  upd(pp.y, gg.y, mm.y, vv.y);
  upd(pp.z, gg.z, mm.z, vv.z);
  upd(pp.w, gg.w, mm.w, vv.w);
  p[o] = pp; m[o] = mm; v[o] = vv;
}

}  // namespace omfs

using namespace omfs;

static GaussW make_gauss() {
  GaussW w;
  double sum = 0.0, g[11];
  for (int i = 0; i < 11; ++i) { g[i] = exp(-((i - 5) * (i - 5)) / (2.0 * 1.5 * 1.5)); sum += g[i]; }
  for (int i = 0; i < 11; ++i) w.g[i] = (float)(g[i] / sum);
  return w;
}

extern "C" int omfs_loss_l1_ssim(const float* image, const float* target, int width, int height, float lambda_dssim,
                                 float* dimage, float* loss_out, float* scratch, void* stream) {
  OMFS_REQUIRE(image && target && dimage && loss_out && scratch && width > 0 && height > 0, "args");
  hipStream_t s = (hipStream_t)stream;
  const GaussW gw = make_gauss();
  const size_t n = (size_t)3 * width * height;
  const float inv = 1.f / (float)n;
  const float w_l1 = (1.f - lambda_dssim) * inv, w_ssim = lambda_dssim * inv;
  float* m0 = scratch; float* m1 = scratch + n; float* m2 = scratch + 2 * n;
  dim3 grid(cdiv(width, LT), cdiv(height, LT), 3);
  // block partials are parked at the head of dimage (overwritten by ssim_bwd afterwards)
  hipLaunchKernelGGL(ssim_fwd_kernel, grid, dim3(256), 0, s, image, target, width, height, w_l1, w_ssim, gw, m0, m1, m2, dimage);
  OMFS_CHECK_HIP(hipGetLastError());
  hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(1024), 0, s, dimage, (int)(grid.x * grid.y * grid.z), lambda_dssim, loss_out);
  hipLaunchKernelGGL(ssim_bwd_kernel, grid, dim3(256), 0, s, image, target, width, height, w_l1, w_ssim, gw, m0, m1, m2, dimage);
  OMFS_CHECK_HIP(hipGetLastError());
  return OMFS_OK;
}

extern "C" int omfs_adam_step(float* params, const float* grads, float* m, float* v, int n, int n_pad,
                              const omfs_adam_params* ap, void* stream) {
  OMFS_REQUIRE(params && grads && m && v && ap, "null pointer");
  OMFS_REQUIRE(n > 0 && n_pad >= n && n_pad % 256 == 0 && ap->step >= 1, "shape");
  AdamK k;
  const double bc1 = 1.0 - pow((double)ap->beta1, ap->step), bc2 = 1.0 - pow((double)ap->beta2, ap->step);
  for (int i = 0; i < OMFS_NPLANES; ++i) k.lr_step[i] = (float)(ap->lr[i] / bc1);
  k.b1 = ap->beta1; k.b2 = ap->beta2; k.eps = ap->eps; k.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  k.grad_scale = ap->grad_scale;
  const int n4 = n_pad / 4;
  hipLaunchKernelGGL(adam_kernel, dim3(cdiv(n4, 256), OMFS_NPLANES), dim3(256), 0, (hipStream_t)stream, (float4*)params,
                     (const float4*)grads, (float4*)m, (float4*)v, n4, k);
  OMFS_CHECK_HIP(hipGetLastError());
  return OMFS_OK;
}

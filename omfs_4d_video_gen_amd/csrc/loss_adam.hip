// Photometric loss (L1 + D-SSIM) forward+backward in image space, and the fused per-Gaussian Adam.
//
// Loss spec: SURVEY.md Appendix A item 8: (1-l) L1 + l (1 - SSIM), SSIM with an 11x11 Gaussian
// window (sigma 1.5), zero padding, per channel, C1 = 0.01^2, C2 = 0.03^2, mean over 3HW.
// Two kernels, each a streaming separable 11-tap convolution (64-column strips, register ring of rows):
//   ssim_fwd : moments -> SSIM value + the three partial-derivative maps (d/dmu1, d/dE[xx], d/dE[xy])
//   ssim_bwd : convolve the maps back, add the L1 sign term -> dL/dimage
// Adam spec: SURVEY.md Appendix A item 9, torch.optim.Adam semantics; one pass over [59][n_pad],
// 16 bytes per lane: reads p,g,m,v, writes p,m,v = 7 x 4 B per element (HBM-streaming).
#include "common.hpp"

namespace omfs {

constexpr int HALO = 5;
constexpr int SW = 64;                 // output columns per wave
#ifndef OMFS_SSIM_ROWS
#define OMFS_SSIM_ROWS 34              // output rows per wave; rows + 10 must be a multiple of 11
#endif
constexpr int SRH = OMFS_SSIM_ROWS;
constexpr int SROWS = SRH + 2 * HALO;  // input rows streamed per wave
static_assert(SROWS % 11 == 0, "the register ring is unrolled by the 11 window taps");
#ifndef OMFS_SSIM_ROWS_BWD
#define OMFS_SSIM_ROWS_BWD OMFS_SSIM_ROWS
#endif
constexpr int SRH_B = OMFS_SSIM_ROWS_BWD;   // the backward kernel's own strip height
constexpr int SROWS_B = SRH_B + 2 * HALO;
static_assert(SROWS_B % 11 == 0, "the register ring is unrolled by the 11 window taps");

#ifndef OMFS_SSIM_PF
#define OMFS_SSIM_PF 3
#endif
constexpr int SSIM_PF = OMFS_SSIM_PF;   // input rows in flight per wave
struct GaussW { float g[11]; };  // normalised 11-tap window, passed by value (scalar registers)

// Which strip a workgroup takes.  Workgroups go round-robin to the 8 XCDs (linear id mod 8), each with an L2 of its own; strips
// that are neighbours in the image share their halo columns and rows, so XCD k takes a contiguous run of the strip order
// (x fastest, then y, then channel): neighbours then meet in ONE L2 instead of fetching the shared bytes twice from memory.
__device__ __forceinline__ void ssim_strip_of_block(int& bx, int& by, int& bz, uint32_t nz = 3) {
#ifndef OMFS_SSIM_NO_XCD_ORDER
  const uint32_t gx = gridDim.x, gy = gridDim.y, n = gx * gy * nz;      // (layers z >= nz, if any, follow in the linear order)
  const uint32_t lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
  const uint32_t xcd = lin & 7u, idx = lin >> 3, chunk = n >> 3, rem = n & 7u;
  const uint32_t s = xcd * chunk + min(xcd, rem) + idx;
  bx = (int)(s % gx); by = (int)((s / gx) % gy); bz = (int)(s / (gx * gy));
#else
  bx = blockIdx.x; by = blockIdx.y; bz = blockIdx.z;
#endif
}

// Packed fp32: one v_pk_fma_f32 / v_pk_mul_f32 does two IEEE operations for ~1.2x the issue time of one (4.7 against 2.7-4
// cycles per wave-instruction, tools/micro/valu_rate.hip), and these kernels are bound by instruction issue, not by HBM
// (17.1 M + 11.0 M VALU wave-instructions per launch at 2.5 TB/s).  The five moment maps of the forward pass and the three
// derivative maps of the backward pass are convolved in pairs; the arithmetic per element is unchanged (same operations,
// same order), so the results are bit-identical to the scalar form.
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 pk_fma(float g, f2 b, f2 c) { return __builtin_elementwise_fma((f2){g, g}, b, c); }

// Streaming separable window: one wave owns a 64-column strip and walks SROWS input rows top to bottom.
// Each row goes through LDS once for the horizontal taps; the vertical taps read a ring of the last 11
// horizontally filtered rows that lives in registers (the row loop is unrolled by 11 so every ring index is
// static). Global loads are issued SSIM_PF rows ahead.
//   Two instantiations.  These kernels run at < 3 waves per SIMD, so a wave's time is the sum of ALL its instructions -- and a
// third of them were scalar: per row, the bounds of the image (row inside? column inside? output row inside?) as 64-bit mask
// arithmetic with an execution-mask branch each, and 64-bit row addresses (r03_b: 4.6 M scalar next to 16.0 M vector instructions
// in the forward, 5.4 M next to 9.3 M in the backward).  Seven strips of eight touch no border: for them (EDGE = false) every
// load and store is unconditional, the halo columns are loaded by all lanes (lanes >= 10 repeat lane 9's address), offsets are
// 32-bit, and what is left per row are two wave-uniform tests on "first / last block of 11 rows".  Same arithmetic, same bits.
template <bool EDGE>
__device__ __forceinline__ float ssim_fwd_strip(const float* __restrict__ ip, const float* __restrict__ gp, int width, int height,
                                                int ox, int oy, int l, float w_l1, float w_ssim, const GaussW& gw,
                                                float* __restrict__ m_mu1, float* __restrict__ m_xx, float* __restrict__ m_xy, f2* sxy) {
  const int xa = ox - HALO + l, xb = ox + SW - HALO + (EDGE ? l : min(l, 2 * HALO - 1)), xo = ox + l;
  const bool ina = !EDGE || (xa >= 0 && xa < width), inb = !EDGE || (l < 2 * HALO && xb < width), ino = !EDGE || xo < width;
  auto load_row = [&](int iy, float (&v)[4]) {
    const int y = oy - HALO + iy;
    if (EDGE) {
      v[0] = v[1] = v[2] = v[3] = 0.f;
      if (iy < SROWS && y >= 0 && y < height) {
        const size_t ro = (size_t)y * width;
        if (ina) { v[0] = ip[ro + xa]; v[1] = gp[ro + xa]; }
        if (inb) { v[2] = ip[ro + xb]; v[3] = gp[ro + xb]; }
      }
    } else {
      const int ro = y * width;
      v[0] = ip[ro + xa]; v[1] = gp[ro + xa]; v[2] = ip[ro + xb]; v[3] = gp[ro + xb];
    }
  };
  f2 ring_m[11], ring_q[11];      // horizontally filtered (x, y) and (x^2, y^2) of the last 11 rows
  float ring_c[11];               // ... and x y
  float cur[4], pre[SSIM_PF][4];          // rows iy + 1 .. iy + SSIM_PF are in flight while row iy is filtered
  load_row(0, cur);
#pragma unroll
  for (int k = 0; k < SSIM_PF; ++k) load_row(1 + k, pre[k]);
  float contrib = 0.f;
  constexpr int NBLK = SROWS / 11;
  for (int blk = 0; blk < NBLK; ++blk) {
    const int base = blk * 11;
    const bool first = blk == 0, last = blk == NBLK - 1;
#pragma unroll
    for (int r = 0; r < 11; ++r) {
      const int iy = base + r;
      sxy[l] = (f2){cur[0], cur[1]};
      if (EDGE) { if (l < 2 * HALO) sxy[SW + l] = (f2){cur[2], cur[3]}; }
      else sxy[SW + l] = (f2){cur[2], cur[3]};            // lanes >= 10 fill slots nobody reads
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 4; ++k) cur[k] = pre[0][k];
#pragma unroll
      for (int q = 0; q + 1 < SSIM_PF; ++q)
#pragma unroll
        for (int k = 0; k < 4; ++k) pre[q][k] = pre[q + 1][k];
      if (EDGE) load_row(iy + 1 + SSIM_PF, pre[SSIM_PF - 1]);
      else if (r + 1 + SSIM_PF < 11 || !last) load_row(iy + 1 + SSIM_PF, pre[SSIM_PF - 1]);      // rows behind the strip are not needed
      f2 am = {0.f, 0.f}, aq = {0.f, 0.f};
      float ac = 0.f;
#pragma unroll
      for (int t = 0; t < 11; ++t) {
        const float g = gw.g[t];
        const f2 v = sxy[l + t];
        am = pk_fma(g, v, am); aq = pk_fma(g, v * v, aq); ac = fma_(g, v.x * v.y, ac);
      }
      ring_m[r] = am; ring_q[r] = aq; ring_c[r] = ac;
      if (EDGE) {
        const int yin = oy - HALO + iy;
        if (iy >= HALO && iy < HALO + SRH && yin < height && ino) { const f2 c = sxy[l + HALO]; contrib += w_l1 * fabsf(c.x - c.y); }
      } else {
        // rows HALO .. HALO + SRH - 1 of the strip are its own: a wave-uniform weight instead of a branch (+ 0 leaves the sum as it is)
        const bool own = (r >= HALO || !first) && (r < HALO + SRH - (SROWS - 11) || !last);
        const f2 c = sxy[l + HALO];
        contrib += (own ? w_l1 : 0.f) * fabsf(c.x - c.y);
      }
      const int yout = oy + iy - 2 * HALO;
      if (EDGE ? (iy >= 2 * HALO && yout < height && ino) : (r >= 2 * HALO || !first)) {
        f2 mu = {0.f, 0.f}, ee = {0.f, 0.f};
        float exy = 0.f;
#pragma unroll
        for (int t = 0; t < 11; ++t) {
          const float g = gw.g[t];
          const int q = (r + 1 + t) % 11;
          mu = pk_fma(g, ring_m[q], mu); ee = pk_fma(g, ring_q[q], ee); exy = fma_(g, ring_c[q], exy);
        }
        const float mu1 = mu.x, mu2 = mu.y, exx = ee.x, eyy = ee.y;
        const float C1 = 0.0001f, C2 = 0.0009f;
        const float s11 = exx - mu1 * mu1, s22 = eyy - mu2 * mu2, s12 = exy - mu1 * mu2;
        const float A1 = 2.f * mu1 * mu2 + C1, A2 = 2.f * s12 + C2, B1 = mu1 * mu1 + mu2 * mu2 + C1, B2 = s11 + s22 + C2;
        // v_rcp_f32 (1 ulp) instead of three correctly rounded divisions: this stage is tolerance-level
        const float iB1 = __builtin_amdgcn_rcpf(B1), iB2 = __builtin_amdgcn_rcpf(B2);
        const float iB = iB1 * iB2;
        const float ssim = A1 * A2 * iB;
        const float o_mu1 = (2.f * mu2 * (A2 - A1)) * iB - 2.f * mu1 * ssim * iB1 + 2.f * mu1 * ssim * iB2;
        if (EDGE) {
          const size_t o = (size_t)yout * width + xo;
          m_mu1[o] = o_mu1; m_xx[o] = -ssim * iB2; m_xy[o] = 2.f * A1 * iB;
        } else {
          const int o = yout * width + xo;
          m_mu1[o] = o_mu1; m_xx[o] = -ssim * iB2; m_xy[o] = 2.f * A1 * iB;
        }
        contrib -= w_ssim * ssim;
      }
    }
  }
  return contrib;
}

#ifdef OMFS_SSIM_WPE
#define OMFS_SSIM_ATTR __attribute__((amdgpu_waves_per_eu(OMFS_SSIM_WPE, 8)))
#else
#define OMFS_SSIM_ATTR
#endif
__global__ __launch_bounds__(64) OMFS_SSIM_ATTR void ssim_fwd_kernel(const float* __restrict__ img, const float* __restrict__ gt,
                                                      int width, int height, float w_l1, float w_ssim, GaussW gw,
                                                      float* __restrict__ map_mu1, float* __restrict__ map_xx,
                                                      float* __restrict__ map_xy, float* __restrict__ partials) {
  __shared__ f2 sxy[2 * SW];       // (image, target) of one input row: 64 + 10 columns (the interior form writes 128)
  int bx, by, ch;
  ssim_strip_of_block(bx, by, ch);
  const int l = threadIdx.x;
  const int ox = bx * SW, oy = by * SRH;
  const size_t plane = (size_t)width * height;
  const float* ip = img + ch * plane;
  const float* gp = gt + ch * plane;
  float* m0 = map_mu1 + ch * plane; float* m1 = map_xx + ch * plane; float* m2 = map_xy + ch * plane;
  // a strip that touches no border of the image (and whose plane fits 32-bit offsets): nothing to test per row
  const bool interior = ox >= HALO && ox + SW + HALO <= width && oy >= HALO && oy + SRH + HALO <= height && plane < (size_t)(1u << 30);
  float contrib = interior ? ssim_fwd_strip<false>(ip, gp, width, height, ox, oy, l, w_l1, w_ssim, gw, m0, m1, m2, sxy)
                           : ssim_fwd_strip<true>(ip, gp, width, height, ox, oy, l, w_l1, w_ssim, gw, m0, m1, m2, sxy);
  contrib = wave_sum_all(contrib);
  // one partial per wave (24k same-address atomics would serialise at ~11 ns each), parked behind the three maps; ONE extra wave
  // of ssim_bwd_kernel adds them up in index order (bitwise reproducible), so the loss needs no launch of its own.  (Letting the
  // last wave of THIS kernel do it -- ticket, two levels to keep the returning atomics apart -- cost more than it saved: every
  // wave then waits for its map stores to drain before it may draw, 42 -> 52 us.)
  if (l == 0) partials[(ch * gridDim.y + by) * gridDim.x + bx] = contrib;
}

// The backward strip: convolves the three derivative maps back (same ring structure) and adds the L1 sign term.  EDGE as above.
template <bool EDGE>
__device__ __forceinline__ void ssim_bwd_strip(const float* __restrict__ ip, const float* __restrict__ gp, int width, int height,
                                               int ox, int oy, int l, float w_l1, float w_ssim, const GaussW& gw,
                                               const float* __restrict__ m_mu1, const float* __restrict__ m_xx,
                                               const float* __restrict__ m_xy, float* __restrict__ dimg, f2* s01, float* s2) {
  const int xa = ox - HALO + l, xb = ox + SW - HALO + (EDGE ? l : min(l, 2 * HALO - 1)), xo = ox + l;
  const bool ina = !EDGE || (xa >= 0 && xa < width), inb = !EDGE || (l < 2 * HALO && xb < width), ino = !EDGE || xo < width;
  auto load_row = [&](int iy, float (&v)[8], bool maps, bool pix) {
    const int y = oy - HALO + iy;
    // the pixel pair of the output row this input row completes two iterations later (row iy - 10)
    const int yo = oy + iy - 2 * HALO;
    if (EDGE) {
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = 0.f;
      if (iy < SROWS_B && y >= 0 && y < height) {
        const size_t ro = (size_t)y * width;
        if (ina) { v[0] = m_mu1[ro + xa]; v[1] = m_xx[ro + xa]; v[2] = m_xy[ro + xa]; }
        if (inb) { v[3] = m_mu1[ro + xb]; v[4] = m_xx[ro + xb]; v[5] = m_xy[ro + xb]; }
      }
      if (iy >= 2 * HALO && iy < SROWS_B && yo < height && ino) { v[6] = ip[(size_t)yo * width + xo]; v[7] = gp[(size_t)yo * width + xo]; }
    } else {
      if (maps) {
        const int ro = y * width;
        v[0] = m_mu1[ro + xa]; v[1] = m_xx[ro + xa]; v[2] = m_xy[ro + xa];
        v[3] = m_mu1[ro + xb]; v[4] = m_xx[ro + xb]; v[5] = m_xy[ro + xb];
      }
      if (pix) { v[6] = ip[yo * width + xo]; v[7] = gp[yo * width + xo]; }
    }
  };
  f2 ring01[11];
  float ring2[11];
  float cur[8], pre[SSIM_PF][8];
#pragma unroll
  for (int k = 0; k < 8; ++k) cur[k] = 0.f;
#pragma unroll
  for (int q = 0; q < SSIM_PF; ++q)
#pragma unroll
    for (int k = 0; k < 8; ++k) pre[q][k] = 0.f;
  load_row(0, cur, true, false);
#pragma unroll
  for (int k = 0; k < SSIM_PF; ++k) load_row(1 + k, pre[k], true, 1 + k >= 2 * HALO);
  constexpr int NBLK = SROWS_B / 11;
  for (int blk = 0; blk < NBLK; ++blk) {
    const int base = blk * 11;
    const bool first = blk == 0, last = blk == NBLK - 1;
#pragma unroll
    for (int r = 0; r < 11; ++r) {
      const int iy = base + r;
      s01[l] = (f2){cur[0], cur[1]}; s2[l] = cur[2];
      if (EDGE) { if (l < 2 * HALO) { s01[SW + l] = (f2){cur[3], cur[4]}; s2[SW + l] = cur[5]; } }
      else { s01[SW + l] = (f2){cur[3], cur[4]}; s2[SW + l] = cur[5]; }          // lanes >= 10 fill slots nobody reads
      __syncthreads();
      const float xv = cur[6], yv = cur[7];
#pragma unroll
      for (int k = 0; k < 8; ++k) cur[k] = pre[0][k];
#pragma unroll
      for (int q = 0; q + 1 < SSIM_PF; ++q)
#pragma unroll
        for (int k = 0; k < 8; ++k) pre[q][k] = pre[q + 1][k];
      if (EDGE) load_row(iy + 1 + SSIM_PF, pre[SSIM_PF - 1], true, true);
      else if (r + 1 + SSIM_PF < 11 || !last)       // rows behind the strip are not needed; the pixel pair from input row 10 on
        load_row(iy + 1 + SSIM_PF, pre[SSIM_PF - 1], true, r + 1 + SSIM_PF >= 2 * HALO || !first);
      f2 a01 = {0.f, 0.f};
      float a2 = 0.f;
#pragma unroll
      for (int t = 0; t < 11; ++t) {
        const float g = gw.g[t];
        a01 = pk_fma(g, s01[l + t], a01); a2 = fma_(g, s2[l + t], a2);
      }
      ring01[r] = a01; ring2[r] = a2;
      const int yout = oy + iy - 2 * HALO;
      if (EDGE ? (iy >= 2 * HALO && yout < height && ino) : (r >= 2 * HALO || !first)) {
        f2 c01 = {0.f, 0.f};
        float c2 = 0.f;
#pragma unroll
        for (int t = 0; t < 11; ++t) {
          const float g = gw.g[t];
          const int q = (r + 1 + t) % 11;
          c01 = pk_fma(g, ring01[q], c01); c2 = fma_(g, ring2[q], c2);
        }
        const float c0 = c01.x, c1 = c01.y;
        const float d = xv - yv;
        const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        const float out = w_l1 * sgn - w_ssim * (c0 + 2.f * xv * c1 + yv * c2);
        if (EDGE) dimg[(size_t)yout * width + xo] = out;
        else dimg[yout * width + xo] = out;
      }
    }
  }
}

__global__ __launch_bounds__(64) OMFS_SSIM_ATTR void ssim_bwd_kernel(const float* __restrict__ img, const float* __restrict__ gt,
                                                      int width, int height, float w_l1, float w_ssim, GaussW gw,
                                                      const float* __restrict__ map_mu1, const float* __restrict__ map_xx,
                                                      const float* __restrict__ map_xy, float* __restrict__ dimage,
                                                      const float* __restrict__ partials, int n_partials, float constant,
                                                      float* __restrict__ loss_out) {
  __shared__ f2 s01[2 * SW];       // (d/dmu1, d/dE[xx]) of one map row: 64 + 10 columns (the interior form writes 128)
  __shared__ float s2[2 * SW];     // d/dE[xy]
  const int l = threadIdx.x;
  if (blockIdx.z == 3) {       // the fourth "channel" of the grid: its first wave is the loss reduction, the others leave at once
    if (blockIdx.x | blockIdx.y) return;
    float acc = 0.f;
    for (int i0 = 0; i0 < n_partials; i0 += 64 * 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int i = i0 + u * 64 + l; v[u] = i < n_partials ? partials[i] : 0.f; }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    acc = wave_sum_all(acc);
    if (l == 0) loss_out[0] = constant + acc;
    return;
  }
  int bx, by, ch;
  ssim_strip_of_block(bx, by, ch);
  const int ox = bx * SW, oy = by * SRH_B;
  const size_t plane = (size_t)width * height;
  const size_t co = ch * plane;
  const bool interior = ox >= HALO && ox + SW + HALO <= width && oy >= HALO && oy + SRH_B + HALO <= height && plane < (size_t)(1u << 30);
  if (interior)
    ssim_bwd_strip<false>(img + co, gt + co, width, height, ox, oy, l, w_l1, w_ssim, gw, map_mu1 + co, map_xx + co, map_xy + co, dimage + co, s01, s2);
  else
    ssim_bwd_strip<true>(img + co, gt + co, width, height, ox, oy, l, w_l1, w_ssim, gw, map_mu1 + co, map_xx + co, map_xy + co, dimage + co, s01, s2);
}

struct AdamK {
  float lr_step[OMFS_NPLANES];  // lr / (1 - b1^t); with a device-resident step state: the plain lr
  float b1, b2, eps, inv_sqrt_bc2, grad_scale;
};

// One thread: the per-iteration scalars of a training step, on the device (see omfs_step_state in the header).
__global__ void step_advance_kernel(omfs_step_state* st, omfs_lr_schedule sch, const int32_t* __restrict__ next_table, int table_len,
                                    int32_t* __restrict__ next_out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const int done = st->step;                    // iterations completed before this one = 0-based index of this one
  if (next_table && table_len > 0) {
    const long long j = (long long)done + 1 - (long long)st->table_base;
    next_out[0] = next_table[j < 0 ? 0 : (j >= table_len ? table_len - 1 : j)];
  }
  const int step = done + 1, fstep = st->flame_step + 1;
  st->step = step; st->flame_step = fstep;
  double lr = (double)sch.lr_init;
  if (sch.lr_init != sch.lr_final && sch.max_steps > 0) {
    const double t = fmin(fmax((double)done / (double)sch.max_steps, 0.0), 1.0);
    lr = exp(log((double)sch.lr_init) * (1.0 - t) + log((double)sch.lr_final) * t);
  }
  st->lr_xyz = (float)lr;
  st->inv_bc1 = (float)(1.0 / (1.0 - pow((double)sch.beta1, (double)step)));
  st->inv_sqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow((double)sch.beta2, (double)step)));
  st->flame_inv_bc1 = (float)(1.0 / (1.0 - pow((double)sch.beta1, (double)fstep)));
  st->flame_inv_sqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow((double)sch.beta2, (double)fstep)));
}

// grid = (n_pad/1024 [float4 x 256], 59)
__global__ __launch_bounds__(256) void adam_kernel(float4* __restrict__ p, const float4* __restrict__ g,
                                                   float4* __restrict__ m, float4* __restrict__ v, int n4_per_plane,
                                                   AdamK k, int plane0, const omfs_step_state* __restrict__ st) {
  const int plane = plane0 + blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4_per_plane) return;
  const size_t o = (size_t)plane * n4_per_plane + i;
  float lr = k.lr_step[plane];
  if (st) {      // device-resident schedule (graph replay): position planes take this iteration's rate, all take its bias corrections
    lr = (plane < OMFS_P_SCALE ? st->lr_xyz : lr) * st->inv_bc1;
    k.inv_sqrt_bc2 = st->inv_sqrt_bc2;
  }
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

// The 45 SH planes of degree >= 1 with the gradient formed in place: grid = (n_pad/1024, 5); block y takes the NINE planes of
// SH coefficients k = 1 + 3 y .. 3 + 3 y (three channels each: planes 14 + 9 y .. 22 + 9 y), so a thread loads the view direction
// and dL/dcolour of its four Gaussians ONCE for nine plane updates (one plane per block re-read them 45 times: 40 bytes per
// element instead of 28 -- measured 84 us for the 45 planes against 56 us through the gradient buffer).
// g = Y_k(dir) * drgb[c] -- the basis expressions are those of project_bwd_kernel, operation for operation (the TUs are built with
// -ffp-contract=off: the same IEEE operations give the same bits), then exactly adam_kernel's update.
__device__ __forceinline__ float sh_basis_k(int kk, float x, float y, float z) {
  constexpr float C1 = 0.4886025119029199f;
  constexpr float C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f, 0.5462742152960396f};
  constexpr float C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                           -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};
  const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yzp = y * z, xzp = x * z;
  switch (kk) {                         // uniform over the block
    case 1: return -C1 * y;
    case 2: return C1 * z;
    case 3: return -C1 * x;
    case 4: return C2[0] * xy;
    case 5: return C2[1] * yzp;
    case 6: return C2[2] * (2.f * zz - xx - yy);
    case 7: return C2[3] * xzp;
    case 8: return C2[4] * (xx - yy);
    case 9: return C3[0] * y * (3.f * xx - yy);
    case 10: return C3[1] * xy * z;
    case 11: return C3[2] * y * (4.f * zz - xx - yy);
    case 12: return C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy);
    case 13: return C3[4] * x * (4.f * zz - xx - yy);
    case 14: return C3[5] * z * (xx - yy);
    default: return C3[6] * x * (xx - 3.f * yy);
  }
}

__global__ __launch_bounds__(256) void adam_sh_rest_kernel(float4* __restrict__ p, const float4* __restrict__ drgb,
                                                           const float4* __restrict__ dir, float4* __restrict__ m, float4* __restrict__ v,
                                                           int n4_per_plane, AdamK k, int ncoef, const float4* __restrict__ g_low) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4_per_plane) return;
  if (blockIdx.y >= 5) {               // (g_low given) blocks 5..18: planes 0..13 from the gradient buffer, exactly adam_kernel
    const int plane = blockIdx.y - 5;
    const size_t o = (size_t)plane * n4_per_plane + i;
    const float lr = k.lr_step[plane];
    float4 pp = p[o], gg = g_low[o], mm = m[o], vv = v[o];
    auto upd0 = [&](float& pe, float ge, float& me, float& ve) {
      ge *= k.grad_scale;
      me = fma_(k.b1, me, (1.f - k.b1) * ge);
      ve = fma_(k.b2, ve, (1.f - k.b2) * ge * ge);
      const float denom = fma_(sqrtf(ve), k.inv_sqrt_bc2, k.eps);
      pe = pe - lr * (me / denom);
    };
    upd0(pp.x, gg.x, mm.x, vv.x); upd0(pp.y, gg.y, mm.y, vv.y); upd0(pp.z, gg.z, mm.z, vv.z); upd0(pp.w, gg.w, mm.w, vv.w);
    p[o] = pp; m[o] = mm; v[o] = vv;
    return;
  }
  const int k0 = 1 + 3 * blockIdx.y;
  const float4 X = dir[i], Y = dir[(size_t)n4_per_plane + i], Z = dir[(size_t)2 * n4_per_plane + i];
  const float4 D[3] = {drgb[i], drgb[(size_t)n4_per_plane + i], drgb[(size_t)2 * n4_per_plane + i]};
  auto upd = [&](float& pe, float ge, float& me, float& ve, float lr) {
    ge *= k.grad_scale;
    me = fma_(k.b1, me, (1.f - k.b1) * ge);
    ve = fma_(k.b2, ve, (1.f - k.b2) * ge * ge);
    const float denom = fma_(sqrtf(ve), k.inv_sqrt_bc2, k.eps);
    pe = pe - lr * (me / denom);
  };
#pragma unroll
  for (int dk = 0; dk < 3; ++dk) {
    const int kk = k0 + dk;
    const bool on = kk < ncoef;
    const float4 B = make_float4(sh_basis_k(kk, X.x, Y.x, Z.x), sh_basis_k(kk, X.y, Y.y, Z.y), sh_basis_k(kk, X.z, Y.z, Z.z),
                                 sh_basis_k(kk, X.w, Y.w, Z.w));
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      const int plane = OMFS_P_SH + 3 * kk + ch;
      const size_t o = (size_t)plane * n4_per_plane + i;
      const float lr = k.lr_step[plane];
      float4 pp = p[o], mm = m[o], vv = v[o];
      upd(pp.x, on ? B.x * D[ch].x : 0.f, mm.x, vv.x, lr);
      upd(pp.y, on ? B.y * D[ch].y : 0.f, mm.y, vv.y, lr);
      upd(pp.z, on ? B.z * D[ch].z : 0.f, mm.z, vv.z, lr);
      upd(pp.w, on ? B.w * D[ch].w : 0.f, mm.w, vv.w, lr);
      p[o] = pp; m[o] = mm; v[o] = vv;
    }
  }
}

// The same update on a flat range [offset, offset + count) of the [59][n_pad] buffers (count, offset multiples of 4):
// a data-parallel rank that owns one contiguous shard of the reduce-scattered gradient.  p / g / m / v point at the
// START of the range; the plane of element offset + i selects the learning rate.
__global__ __launch_bounds__(256) void adam_range_kernel(float4* __restrict__ p, const float4* __restrict__ g, float4* __restrict__ m,
                                                         float4* __restrict__ v, long long offset4, long long count4, int n4_per_plane,
                                                         AdamK k) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count4) return;
  const int plane = (int)((offset4 + i) / n4_per_plane);
  const float lr = k.lr_step[plane];
  float4 pp = p[i], gg = g[i], mm = m[i], vv = v[i];
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
  p[i] = pp; m[i] = mm; v[i] = vv;
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
  dim3 grid(cdiv(width, SW), cdiv(height, SRH), 3);
  // per-wave partials are parked behind the maps; the extra wave (z = 3) of the backward launch reduces them
  const int n_part = (int)(grid.x * grid.y * grid.z);
  OMFS_REQUIRE(n_part <= OMFS_LOSS_TAIL, "image too large for the loss partials behind the maps (OMFS_LOSS_TAIL)");
  float* partials = scratch + 3 * n;
  hipLaunchKernelGGL(ssim_fwd_kernel, grid, dim3(64), 0, s, image, target, width, height, w_l1, w_ssim, gw, m0, m1, m2, partials);
  OMFS_CHECK_HIP(hipGetLastError());
  dim3 grid_b(cdiv(width, SW), cdiv(height, SRH_B), 4);
  hipLaunchKernelGGL(ssim_bwd_kernel, grid_b, dim3(64), 0, s, image, target, width, height, w_l1, w_ssim, gw, m0, m1, m2, dimage,
                     (const float*)partials, n_part, lambda_dssim, loss_out);
  OMFS_CHECK_HIP(hipGetLastError());
  return OMFS_OK;
}

static int adam_launch(float* params, const float* grads, float* m, float* v, int n, int n_pad, const omfs_adam_params* ap,
                       int plane0, int n_planes, const omfs_step_state* state_dev, void* stream) {
  OMFS_REQUIRE(params && grads && m && v && ap, "null pointer");
  OMFS_REQUIRE(n > 0 && n_pad >= n && n_pad % 256 == 0 && (state_dev || ap->step >= 1), "shape");
  OMFS_REQUIRE(plane0 >= 0 && n_planes >= 1 && plane0 + n_planes <= OMFS_NPLANES, "plane range");
  AdamK k;
  if (state_dev) {
    for (int i = 0; i < OMFS_NPLANES; ++i) k.lr_step[i] = ap->lr[i];
    k.inv_sqrt_bc2 = 1.f;
  } else {
    const double bc1 = 1.0 - pow((double)ap->beta1, ap->step), bc2 = 1.0 - pow((double)ap->beta2, ap->step);
    for (int i = 0; i < OMFS_NPLANES; ++i) k.lr_step[i] = (float)(ap->lr[i] / bc1);
    k.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  }
  k.b1 = ap->beta1; k.b2 = ap->beta2; k.eps = ap->eps;
  k.grad_scale = ap->grad_scale;
  const int n4 = n_pad / 4;
  hipLaunchKernelGGL(adam_kernel, dim3(cdiv(n4, 256), n_planes), dim3(256), 0, (hipStream_t)stream, (float4*)params,
                     (const float4*)grads, (float4*)m, (float4*)v, n4, k, plane0, state_dev);
  OMFS_CHECK_HIP(hipGetLastError());
  return OMFS_OK;
}

extern "C" int omfs_adam_step_sh_rest(float* params, const float* grads_low, const float* drgb, const float* dir, float* m, float* v,
                                      int n, int n_pad, const omfs_adam_params* ap, int sh_degree, void* stream) {
  OMFS_REQUIRE(params && drgb && dir && m && v && ap, "null pointer");
  OMFS_REQUIRE(n > 0 && n_pad >= n && n_pad % 256 == 0 && ap->step >= 1 && sh_degree >= 0 && sh_degree <= 3, "shape");
  AdamK k;
  const double bc1 = 1.0 - pow((double)ap->beta1, ap->step), bc2 = 1.0 - pow((double)ap->beta2, ap->step);
  for (int i = 0; i < OMFS_NPLANES; ++i) k.lr_step[i] = (float)(ap->lr[i] / bc1);
  k.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  k.b1 = ap->beta1; k.b2 = ap->beta2; k.eps = ap->eps;
  k.grad_scale = ap->grad_scale;
  const int n4 = n_pad / 4;
  // the five nine-plane blocks first (they take nine times as long as a one-plane block), then planes 0..13 if asked for
  hipLaunchKernelGGL(adam_sh_rest_kernel, dim3(cdiv(n4, 256), grads_low ? 5 + OMFS_P_SH + 3 : 5), dim3(256), 0, (hipStream_t)stream,
                     (float4*)params, (const float4*)drgb, (const float4*)dir, (float4*)m, (float4*)v, n4, k, (sh_degree + 1) * (sh_degree + 1),
                     (const float4*)grads_low);
  OMFS_CHECK_HIP(hipGetLastError());
  return OMFS_OK;
}

extern "C" int omfs_adam_step_range(float* params, const float* grads, float* m, float* v, int n_pad, long long offset,
                                    long long count, const omfs_adam_params* ap, void* stream) {
  OMFS_REQUIRE(params && grads && m && v && ap, "null pointer");
  OMFS_REQUIRE(n_pad > 0 && n_pad % 256 == 0 && ap->step >= 1, "shape");
  OMFS_REQUIRE(offset >= 0 && count > 0 && offset % 4 == 0 && count % 4 == 0 && offset + count <= (long long)OMFS_NPLANES * n_pad, "range");
  AdamK k;
  const double bc1 = 1.0 - pow((double)ap->beta1, ap->step), bc2 = 1.0 - pow((double)ap->beta2, ap->step);
  for (int i = 0; i < OMFS_NPLANES; ++i) k.lr_step[i] = (float)(ap->lr[i] / bc1);
  k.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  k.b1 = ap->beta1; k.b2 = ap->beta2; k.eps = ap->eps; k.grad_scale = ap->grad_scale;
  const long long count4 = count / 4;
  hipLaunchKernelGGL(adam_range_kernel, dim3((unsigned)((count4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (float4*)params,
                     (const float4*)grads, (float4*)m, (float4*)v, offset / 4, count4, n_pad / 4, k);
  OMFS_CHECK_HIP(hipGetLastError());
  return OMFS_OK;
}

extern "C" int omfs_step_advance(omfs_step_state* state_dev, const omfs_lr_schedule* sch, const int32_t* next_table, int table_len,
                                 int32_t* next_out, void* stream) {
  OMFS_REQUIRE(state_dev && sch && sch->beta1 > 0.f && sch->beta1 < 1.f && sch->beta2 > 0.f && sch->beta2 < 1.f, "args");
  OMFS_REQUIRE(!next_table || (table_len > 0 && next_out), "next_table needs table_len and next_out");
  hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state_dev, *sch, next_table, table_len, next_out);
  OMFS_CHECK_HIP(hipGetLastError());
  return OMFS_OK;
}

extern "C" int omfs_adam_step_dev(float* params, const float* grads, float* m, float* v, int n, int n_pad,
                                  const omfs_adam_params* ap, const omfs_step_state* state_dev, int plane0, int n_planes,
                                  void* stream) {
  OMFS_REQUIRE(state_dev, "state_dev");
  return adam_launch(params, grads, m, v, n, n_pad, ap, plane0, n_planes, state_dev, stream);
}

extern "C" int omfs_adam_step(float* params, const float* grads, float* m, float* v, int n, int n_pad,
                              const omfs_adam_params* ap, void* stream) {
  return adam_launch(params, grads, m, v, n, n_pad, ap, 0, OMFS_NPLANES, nullptr, stream);
}

extern "C" int omfs_adam_step_planes(float* params, const float* grads, float* m, float* v, int n, int n_pad,
                                     const omfs_adam_params* ap, int plane0, int n_planes, void* stream) {
  return adam_launch(params, grads, m, v, n, n_pad, ap, plane0, n_planes, nullptr, stream);
}

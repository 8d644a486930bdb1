// Adaptive density control of the mesh-bound Gaussians (what the absent upstream train.py
// does between iterations, call site 02_Visual_Engine/train_ghost.py:227-271): classify every Gaussian as kept /
// cloned / split / pruned, then compact the SoA (parameters, parent triangle, Adam moments) into new arrays
//     [ kept ... | clones ... | first children ... | second children ... ]
// each group in index order.  Three launches: classify (+ per-256 counts), one-workgroup scan of the counts, compact.
// HBM streaming work: 59 planes read once, written once; runs a handful of times per training run.
// The two children of a split Gaussian are samples of the Gaussian itself in its local (triangle-relative) frame,
// drawn with a counter-based generator keyed by (seed, iteration, Gaussian, child, axis): every data-parallel rank
// draws the same samples without any exchange, and oracle/densify_ref.py restates the generator bit for bit.
#include "common.hpp"

namespace omfs {

constexpr int DB = 256;               // Gaussians per workgroup = counting granule
constexpr uint8_t CLS_KEEP = 1, CLS_CLONE = 2, CLS_SPLIT = 4;

__global__ __launch_bounds__(DB) void densify_classify_kernel(int n, int n_pad, const float* __restrict__ params,
                                                              const int32_t* __restrict__ binding,
                                                              const float* __restrict__ face_xf, const float* __restrict__ stats,
                                                              omfs_densify_params dp, uint8_t* __restrict__ cls,
                                                              float* __restrict__ grad_out, uint32_t* __restrict__ block_counts,
                                                              int n_blocks) {
  __shared__ uint32_t cnt[3];
  if (threadIdx.x < 3) cnt[threadIdx.x] = 0u;
  __syncthreads();
  const int i = blockIdx.x * DB + threadIdx.x;
  uint8_t c = 0;
  if (i < n) {
    auto P = [&](int plane) { return params[(size_t)plane * n_pad + i]; };
    const float grad = stats[i] / fmaxf(stats[(size_t)n_pad + i], 1.0f);   // mean view-space positional gradient
    if (grad_out) grad_out[i] = grad;
    const float smax = fmaxf(fmaxf(P(OMFS_P_SCALE + 0), P(OMFS_P_SCALE + 1)), P(OMFS_P_SCALE + 2));
    const float world_max = expf(smax) * face_xf[(size_t)binding[i] * 16 + 12];   // largest axis in world units
    const bool hot = grad >= dp.grad_threshold;
    const bool small = world_max <= dp.size_threshold;
    const bool split = hot && !small, clone = hot && small;
    const float opacity = 1.f / (1.f + expf(-P(OMFS_P_OPACITY)));
    bool prune = split || opacity < dp.min_opacity;
    if (dp.prune_size > 0.f) prune = prune || world_max > dp.prune_size;
    c = (prune ? 0 : CLS_KEEP) | (clone ? CLS_CLONE : 0) | (split ? CLS_SPLIT : 0);
    cls[i] = c;
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const unsigned long long b = __ballot((c >> k) & 1);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(&cnt[k], (uint32_t)__popcll(b));
  }
  __syncthreads();
  if (threadIdx.x < 3) block_counts[(size_t)threadIdx.x * n_blocks + blockIdx.x] = cnt[threadIdx.x];
}

// exclusive scan of the three count rows (one workgroup; n_blocks <= a few thousand), totals[3] = row sums
__global__ __launch_bounds__(1024) void densify_scan_kernel(uint32_t* __restrict__ block_counts, int n_blocks,
                                                            uint32_t* __restrict__ totals) {
  __shared__ uint32_t part[1024];
  for (int k = 0; k < 3; ++k) {
    uint32_t* row = block_counts + (size_t)k * n_blocks;
    const int per = (n_blocks + 1023) / 1024;
    const int b0 = threadIdx.x * per, b1 = min(n_blocks, b0 + per);
    uint32_t s = 0;
    for (int b = b0; b < b1; ++b) s += row[b];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {       // Hillis-Steele inclusive scan
      const uint32_t t = threadIdx.x >= (unsigned)d ? part[threadIdx.x - d] : 0u;
      __syncthreads();
      part[threadIdx.x] += t;
      __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - s;
    for (int b = b0; b < b1; ++b) { const uint32_t v = row[b]; row[b] = run; run += v; }
    if (threadIdx.x == 1023) totals[k] = part[1023];
    __syncthreads();
  }
}

// ---- counter-based generator: two rounds of a 32-bit mix over (key, counter); Box-Muller on top.
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ float normal_sample(uint32_t seed_lo, uint32_t seed_hi, uint32_t id, uint32_t slot) {
  const uint32_t k = mix32(seed_lo ^ mix32(seed_hi + 0x9e3779b9u));
  const uint32_t a = mix32(k ^ mix32(id * 6u + slot));
  const uint32_t b = mix32(a + 0x85ebca6bu);
  const float u1 = ((float)(a >> 8) + 1.0f) * (1.0f / 16777216.0f);      // (0, 1]
  const float u2 = (float)(b >> 8) * (1.0f / 16777216.0f);               // [0, 1)
  return sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
}

__global__ __launch_bounds__(DB) void densify_compact_kernel(int n, int n_pad, const float* __restrict__ params,
                                                             const int32_t* __restrict__ binding,
                                                             const float* __restrict__ adam_m, const float* __restrict__ adam_v,
                                                             const uint8_t* __restrict__ cls,
                                                             const uint32_t* __restrict__ block_offsets, int n_blocks,
                                                             const uint32_t* __restrict__ totals, omfs_densify_params dp,
                                                             int n_out_pad, float* __restrict__ params_out,
                                                             int32_t* __restrict__ binding_out, float* __restrict__ m_out,
                                                             float* __restrict__ v_out) {
  __shared__ uint32_t wave_cnt[3][DB / 64];
  const int i = blockIdx.x * DB + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint8_t c = i < n ? cls[i] : 0;
  uint32_t rank[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const unsigned long long b = __ballot((c >> k) & 1);
    rank[k] = (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
    if (lane == 0) wave_cnt[k][wave] = (uint32_t)__popcll(b);
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    for (int w = 0; w < wave; ++w) rank[k] += wave_cnt[k][w];
    rank[k] += block_offsets[(size_t)k * n_blocks + blockIdx.x];
  }
  if (c == 0) return;
  const uint32_t n_keep = totals[0], n_clone = totals[1], n_split = totals[2];
  const int32_t face = binding[i];
  auto P = [&](int plane) { return params[(size_t)plane * n_pad + i]; };
  if (c & CLS_KEEP) {                       // the Gaussian itself, with its Adam moments
    const size_t o = rank[0];
    for (int p = 0; p < OMFS_NPLANES; ++p) {
      const size_t src = (size_t)p * n_pad + i, dst = (size_t)p * n_out_pad + o;
      params_out[dst] = params[src]; m_out[dst] = adam_m[src]; v_out[dst] = adam_v[src];
    }
    binding_out[o] = face;
  }
  if (c & CLS_CLONE) {                      // a copy with fresh moments (the outputs arrive zeroed)
    const size_t o = (size_t)n_keep + rank[1];
    for (int p = 0; p < OMFS_NPLANES; ++p) params_out[(size_t)p * n_out_pad + o] = params[(size_t)p * n_pad + i];
    binding_out[o] = face;
  }
  if (c & CLS_SPLIT) {                      // two samples of the Gaussian, scales / 1.6
    float qw = P(OMFS_P_ROT + 0), qx = P(OMFS_P_ROT + 1), qy = P(OMFS_P_ROT + 2), qz = P(OMFS_P_ROT + 3);
    const float qn = sqrtf(qw * qw + qx * qx + qy * qy + qz * qz);
    qw /= qn; qx /= qn; qy /= qn; qz /= qn;
    const float R[9] = {1.f - 2.f * (qy * qy + qz * qz), 2.f * (qx * qy - qw * qz), 2.f * (qx * qz + qw * qy),
                        2.f * (qx * qy + qw * qz), 1.f - 2.f * (qx * qx + qz * qz), 2.f * (qy * qz - qw * qx),
                        2.f * (qx * qz - qw * qy), 2.f * (qy * qz + qw * qx), 1.f - 2.f * (qx * qx + qy * qy)};
    const float ls[3] = {P(OMFS_P_SCALE + 0), P(OMFS_P_SCALE + 1), P(OMFS_P_SCALE + 2)};
    const float mean[3] = {P(OMFS_P_XYZ + 0), P(OMFS_P_XYZ + 1), P(OMFS_P_XYZ + 2)};
    for (int child = 0; child < 2; ++child) {
      const size_t o = (size_t)n_keep + n_clone + (size_t)child * n_split + rank[2];
      float d[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) d[a] = normal_sample(dp.seed_lo, dp.seed_hi, (uint32_t)i, (uint32_t)(child * 3 + a)) * expf(ls[a]);
      for (int p = 0; p < OMFS_NPLANES; ++p) {
        float val = params[(size_t)p * n_pad + i];
        if (p >= OMFS_P_XYZ && p < OMFS_P_XYZ + 3) {
          const int r = p - OMFS_P_XYZ;
          val = mean[r] + (R[3 * r] * d[0] + R[3 * r + 1] * d[1] + R[3 * r + 2] * d[2]);
        } else if (p >= OMFS_P_SCALE && p < OMFS_P_SCALE + 3) {
          val = ls[p - OMFS_P_SCALE] - 0.4700036292457356f;     // log(exp(s) / 1.6)
        }
        params_out[(size_t)p * n_out_pad + o] = val;
      }
      binding_out[o] = face;
    }
  }
}

}  // namespace omfs

using namespace omfs;

extern "C" int omfs_densify_classify(const omfs_gaussians* g, const float* face_xf, const float* stats,
                                     const omfs_densify_params* p, uint8_t* cls, float* grad_out, uint32_t* block_counts,
                                     void* stream) {
  OMFS_REQUIRE(g && face_xf && stats && p && cls && block_counts, "null pointer");
  OMFS_REQUIRE(g->n > 0 && g->n_pad >= g->n && g->params && g->binding, "gaussians");
  const int n_blocks = cdiv(g->n, DB);
  hipLaunchKernelGGL(densify_classify_kernel, dim3(n_blocks), dim3(DB), 0, (hipStream_t)stream, g->n, g->n_pad, g->params,
                     g->binding, face_xf, stats, *p, cls, grad_out, block_counts, n_blocks);
  OMFS_CHECK_HIP(hipGetLastError());
  return OMFS_OK;
}

extern "C" int omfs_densify_scan(uint32_t* block_counts, int n, uint32_t* totals, void* stream) {
  OMFS_REQUIRE(block_counts && totals && n > 0, "args");
  hipLaunchKernelGGL(densify_scan_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, block_counts, cdiv(n, DB), totals);
  OMFS_CHECK_HIP(hipGetLastError());
  return OMFS_OK;
}

extern "C" int omfs_densify_compact(const omfs_gaussians* g, const float* adam_m, const float* adam_v, const uint8_t* cls,
                                    const uint32_t* block_offsets, const uint32_t* totals, const omfs_densify_params* p,
                                    int n_out_pad, float* params_out, int32_t* binding_out, float* m_out, float* v_out,
                                    void* stream) {
  OMFS_REQUIRE(g && adam_m && adam_v && cls && block_offsets && totals && p && params_out && binding_out && m_out && v_out,
               "null pointer");
  OMFS_REQUIRE(g->n > 0 && g->n_pad >= g->n && g->params && g->binding && n_out_pad > 0, "sizes");
  const int n_blocks = cdiv(g->n, DB);
  hipLaunchKernelGGL(densify_compact_kernel, dim3(n_blocks), dim3(DB), 0, (hipStream_t)stream, g->n, g->n_pad, g->params,
                     g->binding, adam_m, adam_v, cls, block_offsets, n_blocks, totals, *p, n_out_pad, params_out, binding_out,
                     m_out, v_out);
  OMFS_CHECK_HIP(hipGetLastError());
  return OMFS_OK;
}

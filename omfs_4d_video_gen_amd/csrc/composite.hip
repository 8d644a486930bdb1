// Front-to-back alpha compositing, forward and backward.  One 256-thread workgroup per 16x16 tile
// (wave w owns pixel rows 4w..4w+3), tile lists staged through LDS in batches of 256 splats
// (one coalesced id load + three 16-byte record gathers per lane, then conflict-free broadcast
// reads).  Heavy tiles are dispatched first (tile_order).
//
// Spec: SURVEY.md Appendix A items 6-7: integer pixel coordinate is the sample position;
// power = -0.5(A dx^2 + C dy^2) - B dx dy, skip power > 0; alpha = min(0.99, o exp(power)), skip
// alpha < 1/255; stop a pixel before a splat that would take T below 1e-4; out = C + T bg.
#include "common.hpp"

namespace omfs {

constexpr int CB = 256;  // splats per LDS batch == threads per tile
constexpr int SB = 64;   // backward: splats per reduction sub-batch

struct CompCam {
  int width, height, gx;
  float bg[3];
};

__global__ __launch_bounds__(256) void composite_fwd_kernel(CompCam cam, const uint32_t* __restrict__ tile_order,
                                                            const uint32_t* __restrict__ tile_start,
                                                            const uint32_t* __restrict__ sorted_ids,
                                                            const float4* __restrict__ g0, const float4* __restrict__ g1,
                                                            const float4* __restrict__ g2, float* __restrict__ image,
                                                            float* __restrict__ final_T, uint32_t* __restrict__ n_contrib) {
  __shared__ float4 s0[CB];
  __shared__ float4 s1[CB];
  __shared__ float sb[CB];
  const uint32_t tile = tile_order[blockIdx.x];
  const int tid = threadIdx.x;
  const int tx0 = (tile % cam.gx) * OMFS_TILE, ty0 = (tile / cam.gx) * OMFS_TILE;
  const int px = tx0 + (tid & 15), py = ty0 + (tid >> 4);
  const bool inside = px < cam.width && py < cam.height;
  const float fx = (float)px, fy = (float)py;
  const uint32_t beg = tile_start[tile], end = tile_start[tile + 1];
  float T = 1.f, C0 = 0.f, C1 = 0.f, C2 = 0.f;
  uint32_t contributor = 0, last = 0;
  bool done = !inside;
  for (uint32_t b = beg; b < end; b += CB) {
    if (__syncthreads_count(done) == CB) break;
    const uint32_t k = b + tid;
    if (k < end) {
      const uint32_t id = sorted_ids[k];
      s0[tid] = g0[id];
      s1[tid] = g1[id];
      sb[tid] = g2[id].x;
    }
    __syncthreads();
    const int cnt = (int)min((uint32_t)CB, end - b);
    if (!done) {
      for (int j = 0; j < cnt; ++j) {
        ++contributor;
        const float4 a = s0[j];
        const float4 c = s1[j];
        const float dx = a.x - fx, dy = a.y - fy;
        const float power = fma_(-0.5f, fma_(a.z * dx, dx, c.x * dy * dy), -(a.w * dx) * dy);
        if (power > 0.f) continue;
        const float alpha = fminf(0.99f, c.y * __expf(power));
        if (alpha < (1.f / 255.f)) continue;
        const float Tn = T * (1.f - alpha);
        if (Tn < 1e-4f) { done = true; break; }
        const float w = alpha * T;
        C0 = fma_(c.z, w, C0);
        C1 = fma_(c.w, w, C1);
        C2 = fma_(sb[j], w, C2);
        T = Tn;
        last = contributor;
      }
    }
  }
  if (inside) {
    const size_t plane = (size_t)cam.width * cam.height, o = (size_t)py * cam.width + px;
    image[o] = fma_(T, cam.bg[0], C0);
    image[plane + o] = fma_(T, cam.bg[1], C1);
    image[2 * plane + o] = fma_(T, cam.bg[2], C2);
    final_T[o] = T;
    n_contrib[o] = last;
  }
}

// Backward: same tile/pixel mapping, list walked back to front.  Per splat the 256 pixel
// contributions are reduced with DPP wave sums, the four waves meet in LDS, and one 64-byte
// record per (tile, splat) is added to dsplat with float atomics (16 lanes per record, so every
// atomic wave-instruction is four whole 64-byte segments: MI355X_MICROARCH "Global float atomics").
__global__ __launch_bounds__(256) void composite_bwd_kernel(CompCam cam, const uint32_t* __restrict__ tile_order,
                                                            const uint32_t* __restrict__ tile_start,
                                                            const uint32_t* __restrict__ sorted_ids,
                                                            const float4* __restrict__ g0, const float4* __restrict__ g1,
                                                            const float4* __restrict__ g2,
                                                            const float* __restrict__ final_T,
                                                            const uint32_t* __restrict__ n_contrib,
                                                            const float* __restrict__ dimage, float* __restrict__ dsplat) {
  __shared__ float4 s0[CB];
  __shared__ float4 s1[CB];
  __shared__ float sb[CB];
  __shared__ uint32_t sid[CB];
  __shared__ float red[SB][4][9 + 2];  // [splat in sub-batch][wave][value], padded to 11
  const uint32_t tile = tile_order[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tx0 = (tile % cam.gx) * OMFS_TILE, ty0 = (tile / cam.gx) * OMFS_TILE;
  const int px = tx0 + (tid & 15), py = ty0 + (tid >> 4);
  const bool inside = px < cam.width && py < cam.height;
  const float fx = (float)px, fy = (float)py;
  const uint32_t beg = tile_start[tile], end = tile_start[tile + 1];
  if (beg == end) return;
  const size_t plane = (size_t)cam.width * cam.height, o = (size_t)py * cam.width + px;
  const float T_final = inside ? final_T[o] : 0.f;
  const uint32_t last = inside ? n_contrib[o] : 0u;
  float dL0 = 0.f, dL1 = 0.f, dL2 = 0.f;
  if (inside) { dL0 = dimage[o]; dL1 = dimage[plane + o]; dL2 = dimage[2 * plane + o]; }
  float T = T_final;
  float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f;         // colour accumulated behind the current splat
  float la = 0.f, lc0 = 0.f, lc1 = 0.f, lc2 = 0.f;  // last alpha / colour
  const float bgdot = dL0 * cam.bg[0] + dL1 * cam.bg[1] + dL2 * cam.bg[2];
  // the block's maximum contributor bounds how much of the list has to be visited
  __shared__ uint32_t s_max;
  if (tid == 0) s_max = 0;
  __syncthreads();
  atomicMax(&s_max, last);
  __syncthreads();
  const uint32_t n_visit = s_max;
  if (n_visit == 0) return;
  const uint32_t n_batches = (n_visit + CB - 1) / CB;
  for (int bi = (int)n_batches - 1; bi >= 0; --bi) {
    const uint32_t b = beg + (uint32_t)bi * CB;
    const int cnt = (int)min((uint32_t)CB, n_visit - (uint32_t)bi * CB);
    __syncthreads();
    if (tid < cnt) {
      const uint32_t id = sorted_ids[b + tid];
      sid[tid] = id;
      s0[tid] = g0[id];
      s1[tid] = g1[id];
      sb[tid] = g2[id].x;
    }
    __syncthreads();
    for (int sbase = ((cnt - 1) / SB) * SB; sbase >= 0; sbase -= SB) {
      const int scnt = min(SB, cnt - sbase);
      for (int jj = scnt - 1; jj >= 0; --jj) {
        const int j = sbase + jj;
        const uint32_t contributor = (uint32_t)bi * CB + (uint32_t)j + 1u;  // 1-based position in the list
        const bool mine = contributor <= last;
        if (__ballot(mine) == 0ull) {  // wave-uniform: nothing behind this splat for any of our pixels
          if (lane == 63) {
#pragma unroll
            for (int q = 0; q < 9; ++q) red[jj][wave][q] = 0.f;
          }
          continue;
        }
        float v[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) v[q] = 0.f;
        if (mine) {
          const float4 a = s0[j];
          const float4 c = s1[j];
          const float dx = a.x - fx, dy = a.y - fy;
          const float power = fma_(-0.5f, fma_(a.z * dx, dx, c.x * dy * dy), -(a.w * dx) * dy);
          const float G = __expf(power);
          const float alpha = fminf(0.99f, c.y * G);
          if (power <= 0.f && alpha >= (1.f / 255.f)) {
            T = T / (1.f - alpha);
            const float w = alpha * T;
            const float cb = sb[j];
            v[6] = w * dL0; v[7] = w * dL1; v[8] = w * dL2;  // dL/dcolour
            acc0 = fma_(la, lc0, (1.f - la) * acc0);
            acc1 = fma_(la, lc1, (1.f - la) * acc1);
            acc2 = fma_(la, lc2, (1.f - la) * acc2);
            lc0 = c.z; lc1 = c.w; lc2 = cb; la = alpha;
            float dLa = (c.z - acc0) * dL0 + (c.w - acc1) * dL1 + (cb - acc2) * dL2;
            dLa = dLa * T + (-T_final / (1.f - alpha)) * bgdot;
            // alpha = min(0.99, o*G) is differentiated straight through the clamp, as the upstream
            // rasteriser does (DESIGN.md "Frozen conventions").
            const float dLG = c.y * dLa, gdx = G * dx, gdy = G * dy;
            v[0] = dLG * (-gdx * a.z - gdy * a.w);  // d mean2d.x (dx = mean - pixel)
            v[1] = dLG * (-gdy * c.x - gdx * a.w);  // d mean2d.y
            v[2] = -0.5f * gdx * dx * dLG;          // d conic.a
            v[3] = -gdx * dy * dLG;                 // d conic.b (B multiplies dx*dy once)
            v[4] = -0.5f * gdy * dy * dLG;          // d conic.c
            v[5] = G * dLa;                         // d opacity
          }
        }
#pragma unroll
        for (int q = 0; q < 9; ++q) v[q] = wave_sum_to_lane63(v[q]);
        if (lane == 63) {
#pragma unroll
          for (int q = 0; q < 9; ++q) red[jj][wave][q] = v[q];
        }
      }
      __syncthreads();
      // 16 lanes per 64-byte splat record; lane q < 9 adds value q
      for (int r = tid; r < scnt * 16; r += CB) {
        const int jj = r >> 4, q = r & 15;
        if (q < 9) {
          const float sum = (red[jj][0][q] + red[jj][1][q]) + (red[jj][2][q] + red[jj][3][q]);
          if (sum != 0.f) atomicAdd(&dsplat[(size_t)sid[sbase + jj] * 16 + q], sum);
        }
      }
      __syncthreads();
    }
  }
}

__global__ void image_to_rgb8_kernel(const float* __restrict__ image, int width, int height, uint8_t* __restrict__ rgb8) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = width * height;
  if (i >= n) return;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float v = fminf(fmaxf(image[(size_t)c * n + i], 0.f), 1.f);
    rgb8[(size_t)i * 3 + c] = (uint8_t)(v * 255.f + 0.5f);
  }
}

static CompCam make_compcam(const omfs_camera* c) {
  CompCam k;
  k.width = c->width; k.height = c->height; k.gx = cdiv(c->width, OMFS_TILE);
  for (int i = 0; i < 3; ++i) k.bg[i] = c->bg[i];
  return k;
}

}  // namespace omfs

using namespace omfs;

extern "C" int omfs_composite_fwd(const omfs_camera* cam, const omfs_raster_buffers* rb, void* stream) {
  OMFS_REQUIRE(cam && rb, "null pointer");
  OMFS_REQUIRE(rb->g0 && rb->g1 && rb->g2 && rb->tile_order && rb->tile_start && rb->sorted_ids && rb->image &&
                   rb->final_T && rb->n_contrib, "raster buffers");
  CompCam cc = make_compcam(cam);
  const int n_tiles = cc.gx * cdiv(cam->height, OMFS_TILE);
  hipLaunchKernelGGL(composite_fwd_kernel, dim3(n_tiles), dim3(256), 0, (hipStream_t)stream, cc, rb->tile_order,
                     rb->tile_start, rb->sorted_ids, (const float4*)rb->g0, (const float4*)rb->g1,
                     (const float4*)rb->g2, rb->image, rb->final_T, rb->n_contrib);
  OMFS_CHECK_HIP(hipGetLastError());
  return OMFS_OK;
}

extern "C" int omfs_composite_bwd(const omfs_camera* cam, const omfs_raster_buffers* rb, const omfs_grad_buffers* gb,
                                  void* stream) {
  OMFS_REQUIRE(cam && rb && gb, "null pointer");
  OMFS_REQUIRE(rb->g0 && rb->g1 && rb->g2 && rb->tile_order && rb->tile_start && rb->sorted_ids && rb->final_T &&
                   rb->n_contrib && gb->dimage && gb->dsplat, "buffers");
  CompCam cc = make_compcam(cam);
  const int n_tiles = cc.gx * cdiv(cam->height, OMFS_TILE);
  hipLaunchKernelGGL(composite_bwd_kernel, dim3(n_tiles), dim3(256), 0, (hipStream_t)stream, cc, rb->tile_order,
                     rb->tile_start, rb->sorted_ids, (const float4*)rb->g0, (const float4*)rb->g1,
                     (const float4*)rb->g2, rb->final_T, rb->n_contrib, gb->dimage, gb->dsplat);
  OMFS_CHECK_HIP(hipGetLastError());
  return OMFS_OK;
}

extern "C" int omfs_image_to_rgb8(const float* image, int width, int height, uint8_t* rgb8, void* stream) {
  OMFS_REQUIRE(image && rgb8 && width > 0 && height > 0, "args");
  hipLaunchKernelGGL(image_to_rgb8_kernel, dim3(cdiv(width * height, 256)), dim3(256), 0, (hipStream_t)stream, image,
                     width, height, rgb8);
  OMFS_CHECK_HIP(hipGetLastError());
  return OMFS_OK;
}

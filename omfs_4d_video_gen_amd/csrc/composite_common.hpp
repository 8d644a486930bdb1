// Shared by composite.hip (the product kernels) and composite_experiments.hip (second implementations of the backward pass,
// built into a library of their own: libomfs_experiments.so): constants, the per-splat sub-block coverage test, the
// workgroup -> (unit, quadrant) dealing, the scalar bit scan, the segment -> tile lookup and the tables the forward pass leaves
// for the backward pass.
#pragma once
#include "common.hpp"

namespace omfs {

constexpr float LOG2E = 1.4426950408889634f;
constexpr float LOG2_INV255 = -7.994353436858858f;  // log2(1/255)

struct CompCam {
  int width, height, gx;
  float bg[3];
};

// 4-bit coverage mask of one splat over the four 4x4-pixel sub-blocks of ONE 8x8 quadrant: bit
// s = (kx&1) + 2(ky&1).  A bit is set when the ellipse {q(d) <= 2 ln(255 o)} -- outside of which
// alpha < 1/255 -- can overlap the sub-block.  Conservative by construction (blocks are widened by half a
// pixel, the ellipse by a rounding slack): a cleared bit means no pixel of the sub-block can receive a
// contribution, so skipping the splat for it changes nothing.  The ellipse is cut by the 3 horizontal
// lines that bound the quadrant's 2 block rows (one sqrt each); inside a row band the x-extent of the
// convex set is attained on the two lines or at the ellipse's leftmost / rightmost point.
__device__ __forceinline__ uint32_t quadrant_mask(float mx, float my, float A, float B, float C, float lo, int qx0, int qy0) {
  const float qmax = 2.f * 0.6931471805599453f * (lo - LOG2_INV255);
  if (!(qmax >= 0.f)) return 0u;            // opacity below 1/255: never contributes
  const float det = A * C - B * B;
  if (!(A > 0.f && C > 0.f && det > 0.f)) return 0xFu;   // degenerate conic: never cull
  const float Q = qmax * 1.0002f + 0.02f;
  const float idet = __builtin_amdgcn_rcpf(det), iA = __builtin_amdgcn_rcpf(A);
  // v_sqrt_f32 (1 ulp): the slack factors below are three orders of magnitude larger than its error
  const float vmax = __builtin_amdgcn_sqrtf(Q * A * idet) * 1.0001f, umax = __builtin_amdgcn_sqrtf(Q * C * idet) * 1.0001f;
  const float vl = B * umax * __builtin_amdgcn_rcpf(C);   // v of the leftmost point (u = -umax); the rightmost is at -vl
  float vline[3], ulo[3], uhi[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    vline[k] = ((float)qy0 - 0.5f + 4.f * (float)k) - my;
    const float vc = fminf(fmaxf(vline[k], -vmax), vmax);
    const float sq = __builtin_amdgcn_sqrtf(fmaxf(A * Q - det * vc * vc, 0.f));
    ulo[k] = (-B * vc - sq) * iA;
    uhi[k] = (-B * vc + sq) * iA;
  }
  const float x0 = ((float)qx0 - 0.5f) - mx;   // left edge of block column 0, relative to the splat centre
  uint32_t m = 0;
#pragma unroll
  for (int ky = 0; ky < 2; ++ky) {
    const float v0 = vline[ky], v1 = vline[ky + 1];
    if (v0 > vmax || v1 < -vmax) continue;
    const float a = fmaxf(v0, -vmax), b = fminf(v1, vmax);
    float xlo = (a <= vl && vl <= b) ? -umax : fminf(ulo[ky], ulo[ky + 1]);
    float xhi = (a <= -vl && -vl <= b) ? umax : fmaxf(uhi[ky], uhi[ky + 1]);
    xlo -= 0.02f + 1e-4f * fabsf(xlo);
    xhi += 0.02f + 1e-4f * fabsf(xhi);
#pragma unroll
    for (int kx = 0; kx < 2; ++kx) {
      const float L = x0 + 4.f * (float)kx;
      if (xhi >= L && xlo <= L + 4.f) m |= 1u << (kx + 2 * ky);
    }
  }
  return m;
}

#include "composite_debug.hpp"

// Which (unit, quadrant) a workgroup takes, unit = tile position or list segment.  Workgroups go round-robin to the 8 XCDs
// (linear id mod 8), each with its own L2, and the four quadrants of a unit gather the SAME list entries and records: with
// (unit, quadrant) = (id / 4, id % 4) the four land on four XCDs and every record is fetched into four L2s.  Here a group of
// 32 R consecutive workgroups takes 8 R consecutive units, XCD k the run of R units k R .. k R + R - 1 of the group with all
// four quadrants of each (the grid is a multiple of 4; a last partial group keeps the plain form: a bijection either way).
#ifndef OMFS_XCD_RUN
#define OMFS_XCD_RUN 1
#endif
__device__ __forceinline__ void unit_quadrant_of_block(uint32_t& unit, int& quad) {
#ifndef OMFS_NO_XCD_ORDER
  constexpr uint32_t R = OMFS_XCD_RUN, G = 32u * R;
  const uint32_t lin = blockIdx.x, group = lin / G;
  if ((group + 1) * G <= gridDim.x) {
    const uint32_t in = lin - group * G, k = in & 7u, a = in >> 3;       // a = 0 .. 4 R - 1 on XCD k
    unit = (group * 8u + k) * R + (a >> 2);
    quad = (int)(a & 3u);
    return;
  }
#endif
  unit = blockIdx.x >> 2;
  quad = (int)(blockIdx.x & 3u);
}

constexpr int WB = 64;   // splats staged per wave and step

// Lowest set bit of a wave-uniform 64-bit mask: returns its index + 1 (0 for an empty mask) and clears it.  Two scalar
// instructions (s_ff1_i32_b64 yields -1 for an empty mask, s_bitset0_b64 then clears bit 63 of a mask that is already 0)
// where `ffsll(m); m &= m - 1` compiles to seven: the scalar unit retires one instruction per ~4 cycles and SIMD
// (tools/micro/valu_rate.hip), so the bit scans of the visit loops are not free.
__device__ __forceinline__ int pop_lowest_bit(unsigned long long& m) {
  int i;
  asm("s_ff1_i32_b64 %0, %1" : "=s"(i) : "s"(m));
  asm("s_bitset0_b64 %0, %1" : "+s"(m) : "s"(i));
  return i + 1;
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, d, 64));
  return v;
}

// (tile, segment within the tile's list) of global list segment `seg`: one load from the table the forward pass left
// (composite_fwd_kernel), or -- no table, or the sentinel of an over-long list -- the launch-order position p with
// order_seg0[p] <= seg < order_seg0[p+1] by bisection (~13 dependent L2-resident loads).
__device__ __forceinline__ void segment_tile(uint32_t seg, int n_tiles, const uint32_t* __restrict__ tile_order,
                                             const uint32_t* __restrict__ order_seg0, const uint32_t* __restrict__ seg_table,
                                             uint32_t& tile, uint32_t& kseg) {
  if (seg_table) {
    const uint32_t e = seg_table[seg];
    if (e != 0xFFFFFFFFu) { tile = e & 0xFFFFu; kseg = e >> 16; return; }
  }
  int lo = 0, hi = n_tiles;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (order_seg0[mid] <= seg) lo = mid; else hi = mid;
  }
  tile = tile_order[lo];
  kseg = seg - order_seg0[lo];
}

// The segment table of the backward pass lives in `keys`: the (depth, id) pairs are dead once omfs_tile_sort has produced
// sorted_ids, and the next frame's binning rewrites them.  2 * dup_capacity words hold seg_capacity entries unless the pair
// capacity is tiny against the tile count (then: no table, the backward bisects).
static inline uint32_t* segment_table(const omfs_raster_buffers* rb) {
  return (rb->keys && rb->order_seg0 && 2ull * rb->dup_capacity >= (unsigned long long)rb->seg_capacity) ? rb->keys : nullptr;
}
// ... followed by one word per (tile, quadrant): the quadrant's depth (see composite_fwd_kernel)
static inline uint32_t* quadrant_depths(const omfs_raster_buffers* rb, int n_tiles) {
  if (rb->quad_depth) return rb->quad_depth;       // the caller's own table (per view: also the forward's priority hint)
  return (segment_table(rb) && 2ull * rb->dup_capacity >= (unsigned long long)rb->seg_capacity + 4ull * (unsigned long long)n_tiles)
             ? rb->keys + rb->seg_capacity : nullptr;
}

static inline CompCam make_compcam(const omfs_camera* c) {
  CompCam k;
  k.width = c->width; k.height = c->height; k.gx = cdiv(c->width, OMFS_TILE);
  for (int i = 0; i < 3; ++i) k.bg[i] = c->bg[i];
  return k;
}

}  // namespace omfs

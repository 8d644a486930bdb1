// Tile binning: scan of per-tile hit counts, key scatter, per-tile radix depth sort in LDS.
//
// Design (DESIGN.md "Binning"): instead of the upstream single global 64-bit radix sort of
// (tile<<32 | depth) keys the D (Gaussian,tile) pairs -- the tiles of
// each 3-sigma rectangle that pass the frozen tile_touched() test -- are counting-sorted by tile
// (LDS-aggregated counts, offsets from one small scan, placement by a per-tile cursor) and each tile's segment is then sorted on its own by a 4x8-bit
// LSD radix sort that lives entirely in LDS (160 KB per CU on gfx950): wave-level digit matching
// (ballots) gives stable ranks, per-wave digit tables give the offsets.  Arrival order inside a
// tile is arbitrary (atomics), so equal depths are finally ordered by Gaussian id: the result is
// exactly the order of the upstream stable global sort whose emission order is the Gaussian index.
// HBM traffic: write 8 B + read 8 B + write 4 B per pair, versus ~24 B x 6 passes for the global sort.
#include <type_traits>
#include "common.hpp"

namespace omfs {

// ------------------------------------------------------------------ scan + launch order
// One workgroup.  Thread t owns the SCAN_PER consecutive tiles starting at t * per; their counts stay in registers:
//   1. exclusive scan of the counts -> tile_start, tile_cursor = 0, overflow flag;
//   2. launch order: tiles sorted by descending log2 bucket of their count (heavy tiles first).  One 64-bit LDS
//      atomic per tile hands out the position inside the bucket (low word) together with the number of
//      OMFS_SEG-entry list segments of the tiles in front of it (high word), so
//   3. order_seg0[p] = segments owned by the tiles before position p of the launch order (order_seg0[n_tiles] =
//      total) needs no further pass.  The backward pass launches one wave per (segment, quadrant) and finds its
//      tile by bisection in this array.
// What this kernel costs is what ONE compute unit can issue: three replacements of the per-tile LDS atomics (wave-
// aggregated counting, a stable sort by packed class counters, the same across several workgroups with a flag exchange)
// all came out SLOWER (31-43 us against 23) because they spend more instructions per tile.  What did pay is the memory
// side: with thread t owning 8 consecutive tiles a direct global access touches 32 cache lines per wave instruction, so
// images of up to 8192 tiles move everything through LDS -- coalesced global <-> LDS, strided / scattered LDS <-> registers.
constexpr int SCAN_PER = 8;   // tiles per thread held in registers (n_tiles <= 8192; larger images loop)

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, true);
}
// inclusive prefix sum over the 64 lanes, DPP only (no LDS round trips)
__device__ __forceinline__ uint32_t wave_incl_scan_u32_dpp(uint32_t v) {
  v += dpp_u32<0x111, 0xF>(v);   // row_shr:1
  v += dpp_u32<0x112, 0xF>(v);   // row_shr:2
  v += dpp_u32<0x114, 0xF>(v);   // row_shr:4
  v += dpp_u32<0x118, 0xF>(v);   // row_shr:8   -> inclusive inside each row of 16
  v += dpp_u32<0x142, 0xA>(v);   // row_bcast:15 -> rows 1, 3
  v += dpp_u32<0x143, 0xC>(v);   // row_bcast:31 -> rows 2, 3
  return v;
}

__global__ __launch_bounds__(1024) void tile_scan_kernel(int n_tiles, uint32_t* __restrict__ tile_count,
                                                         uint32_t* __restrict__ tile_start,
                                                         uint32_t* __restrict__ tile_cursor,
                                                         uint32_t* __restrict__ tile_order, uint32_t dup_capacity,
                                                         uint32_t* __restrict__ status, uint32_t* __restrict__ order_seg0) {
  __shared__ uint32_t wave_tot[16];
  __shared__ unsigned long long bucket_acc[33];   // low: tiles in the bucket, high: their segments; then running bases
  __shared__ uint32_t carry;                      // tiles / pairs of the chunks before (images with > 8192 tiles)
  __shared__ uint32_t s_nonempty, s_total_segs, carry_e;   // empty tiles take no atomic: they follow the others in index order
  extern __shared__ __attribute__((aligned(16))) uint32_t scan_lds[];
  uint32_t* s_a = scan_lds;                       // counts, then tile_start
  uint32_t* s_b = scan_lds + 1024 * SCAN_PER;     // tile_order by position
  uint32_t* s_c = scan_lds + 2 * 1024 * SCAN_PER; // order_seg0 by position
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < 33) bucket_acc[tid] = 0ull;
  if (tid == 0) carry = 0;
  const bool single = n_tiles <= 1024 * SCAN_PER;   // then everything is staged through LDS and cnt[] survives into pass B
  if (single) {
    for (int i = tid; i < 1024 * SCAN_PER; i += 1024) {
      const bool in = i < n_tiles;
      s_a[i] = in ? tile_count[i] : 0u;
      if (in) { tile_count[i] = 0u; tile_cursor[i] = 0u; }     // consumed / reset: coalesced
    }
  }
  __syncthreads();
  // ---- pass A over chunks of 8192 tiles: scan + bucket totals
  uint32_t cnt[SCAN_PER];
  for (int c0 = 0; c0 < n_tiles; c0 += 1024 * SCAN_PER) {
    const int beg = c0 + tid * SCAN_PER;
    if (single) {
      const uint4 a = *reinterpret_cast<const uint4*>(s_a + beg), b = *reinterpret_cast<const uint4*>(s_a + beg + 4);
      cnt[0] = a.x; cnt[1] = a.y; cnt[2] = a.z; cnt[3] = a.w; cnt[4] = b.x; cnt[5] = b.y; cnt[6] = b.z; cnt[7] = b.w;
    } else {
#pragma unroll
      for (int k = 0; k < SCAN_PER; ++k) cnt[k] = beg + k < n_tiles ? tile_count[beg + k] : 0u;
    }
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < SCAN_PER; ++k) sum += cnt[k];
    const uint32_t incl = wave_incl_scan_u32_dpp(sum);
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    uint32_t base = carry, chunk_total = 0;
    for (int w = 0; w < 16; ++w) {
      const uint32_t v = wave_tot[w];
      if (w < wave) base += v;
      chunk_total += v;
    }
    uint32_t run = base + incl - sum;
    uint32_t st[SCAN_PER];
#pragma unroll
    for (int k = 0; k < SCAN_PER; ++k) {
      st[k] = run;
      if (beg + k < n_tiles) {
        if (!single) { tile_start[beg + k] = run; tile_cursor[beg + k] = 0; }   // rewritten as 0 below if the capacity overflows
        const uint32_t c = cnt[k];
        // empty tiles (more than half of them at 1080p) are not counted here: all lanes hitting the one counter of the empty
        // class serialised in the LDS atomic unit, and their place in the launch order needs no counter (pass B)
        if (c) atomicAdd(&bucket_acc[32 - (32 - __clz(c))], 1ull | ((unsigned long long)((c + OMFS_SEG - 1) / OMFS_SEG) << 32));
        run += c;
      }
    }
    __syncthreads();                               // everybody has read its counts from s_a
    if (single) {
      *reinterpret_cast<uint4*>(s_a + beg) = make_uint4(st[0], st[1], st[2], st[3]);
      *reinterpret_cast<uint4*>(s_a + beg + 4) = make_uint4(st[4], st[5], st[6], st[7]);
    }
    if (tid == 0) carry += chunk_total;
    __syncthreads();
  }
  const uint32_t total = carry;
  const bool overflow = total > dup_capacity;
  if (single)
    for (int i = tid; i < n_tiles; i += 1024) tile_start[i] = overflow ? 0u : s_a[i];     // coalesced
  if (tid == 0) {
    tile_start[n_tiles] = overflow ? 0u : total;
    if (overflow) atomicOr(status, OMFS_STATUS_DUP_OVERFLOW);
    unsigned long long r = 0ull;   // exclusive scan over the buckets, both words at once
    for (int b = 0; b < 33; ++b) {
      const unsigned long long v = overflow ? 0ull : bucket_acc[b];   // on overflow every list is emptied: one bucket
      bucket_acc[b] = r;
      r += v;
    }
    order_seg0[n_tiles] = overflow ? 0u : (uint32_t)(r >> 32);
    s_nonempty = (uint32_t)r;                       // tiles with a list: the empty ones are placed behind them
    s_total_segs = overflow ? 0u : (uint32_t)(r >> 32);
    carry_e = 0u;
  }
  __syncthreads();
  // ---- pass B: positions in the launch order (+ segment prefix); counts are re-read only for images with > 8192 tiles
  const uint32_t n_nonempty = s_nonempty, total_segs = s_total_segs;
  for (int c0 = 0; c0 < n_tiles; c0 += 1024 * SCAN_PER) {
    const int beg = c0 + tid * SCAN_PER;
    uint32_t cb[SCAN_PER], n_e = 0;
#pragma unroll
    for (int k = 0; k < SCAN_PER; ++k) {
      cb[k] = 0u;
      if (beg + k < n_tiles) {
        cb[k] = single ? cnt[k] : tile_count[beg + k];
        if (!single) tile_count[beg + k] = 0;          // consumed: the next frame's omfs_bin_count accumulates from zero
        if (overflow) { cb[k] = 0; if (!single) tile_start[beg + k] = 0; }
        n_e += cb[k] == 0u;
      }
    }
    // rank of this thread's first empty tile among the empty tiles (index order): one more block scan per chunk
    const uint32_t incl_e = wave_incl_scan_u32_dpp(n_e);
    if (lane == 63) wave_tot[wave] = incl_e;
    __syncthreads();
    uint32_t e_rank = carry_e + incl_e - n_e, chunk_e = 0;
    for (int w = 0; w < 16; ++w) {
      const uint32_t v = wave_tot[w];
      if (w < wave) e_rank += v;
      chunk_e += v;
    }
#pragma unroll
    for (int k = 0; k < SCAN_PER; ++k) {
      if (beg + k < n_tiles) {
        const uint32_t c = cb[k];
        uint32_t pos, seg;
        if (c) {
          const unsigned long long old = atomicAdd(&bucket_acc[32 - (32 - __clz(c))], 1ull | ((unsigned long long)((c + OMFS_SEG - 1) / OMFS_SEG) << 32));
          pos = (uint32_t)old; seg = (uint32_t)(old >> 32);
        } else {
          pos = n_nonempty + e_rank++; seg = total_segs;
        }
        if (single) { s_b[pos] = (uint32_t)(beg + k); s_c[pos] = seg; }
        else { tile_order[pos] = (uint32_t)(beg + k); order_seg0[pos] = seg; }
      }
    }
    __syncthreads();                               // wave_tot and carry_e are reused by the next chunk
    if (tid == 0) carry_e += chunk_e;
    __syncthreads();
  }
  if (single) {
    __syncthreads();
    for (int i = tid; i < n_tiles; i += 1024) { tile_order[i] = s_b[i]; order_seg0[i] = s_c[i]; }
  }
}

// ------------------------------------------------------------------ count + key scatter
// Both walk every Gaussian's tile rectangle, apply the frozen tile_touched() test, and aggregate
// in LDS first (one counter per tile: 32 KB at 1080p): a 512-Gaussian block produces ~6k
// (Gaussian,tile) pairs but touches <= n_tiles counters, so the global atomics drop from one per
// pair to one per non-empty (block, tile) -- an order of magnitude fewer, and only the scatter's
// are returning atomics.  Placement inside a tile segment is arbitrary; the sort fixes the order.
//
// Load balance: rectangles range from 1 to >100 tiles, so a lane-per-Gaussian loop runs at the pace of the
// largest rectangle in each wave (measured 4-5x the mean).  Instead every wave publishes its 64 rectangles
// in LDS, prefix-sums their areas and walks the CONCATENATED list of rectangle tiles 64 at a time: lane l
// takes flat index p = 64 k + l, finds the owning Gaussian by a 6-step bisection of the prefix sums and
// tests that one tile.  Adjacent lanes mostly test adjacent tiles of the same Gaussian, so the LDS counter
// atomics rarely collide.
#ifndef OMFS_BIN_THREADS
#define OMFS_BIN_THREADS 512
#endif
constexpr int BIN_THREADS = OMFS_BIN_THREADS;
constexpr int BIN_WAVES = BIN_THREADS / 64;

struct PairSource {
  const float4* g0; const float4* g1; const float4* g2;
};

// QUOT: r1 also carries the tile test's two per-Gaussian quotients (-b / c, -b / a), evaluated ONCE per Gaussian by the publishing
// lane instead of once per rectangle tile (two correctly rounded divisions, ~20 of the test's ~200 instructions; same operands,
// same bits).  Only the count kernel evaluates the test for every tile (the scatter replays its ballots), and only there the 8 extra
// bytes per lane fit: its per-tile counters are 16-bit (a workgroup holds 512 Gaussians), which keeps three workgroups -- the whole
// grid -- resident per CU.  (Round 2 published the quotients in both kernels with 32-bit counters: one workgroup per CU fewer, slower.)
template <bool QUOT>
struct WaveRectsT {           // one wave's published rectangles (36 / 44 B per lane)
  float4 r0[64];              // mean2d.xy, conic a, b
  typename std::conditional<QUOT, float4, float2>::type r1[64];   // conic c, opacity (, -b / c, -b / a)
  uint32_t scan[64];          // inclusive prefix sum of rectangle areas
  uint32_t rect[64];          // packed x0 | y0<<8 | x1<<16 | y1<<24
  uint32_t depth[64];         // depth bits
  uint32_t coarse[8];         // scan[7], scan[15], ..., scan[63]: first level of the owner search
};
using WaveRects = WaveRectsT<false>;
constexpr size_t BIN_SCRATCH_BYTES = sizeof(WaveRects) * BIN_WAVES;
constexpr size_t BIN_COUNT_SCRATCH_BYTES = sizeof(WaveRectsT<true>) * BIN_WAVES;

// Which Gaussian lane l of this wave owns (bin_count_kernel and bin_scatter_kernel, which must agree: the scatter replays the
// count's ballots).  The Gaussians are stored in mesh order, so neighbours have similar footprints, and with 512 consecutive
// Gaussians per workgroup / 64 per wave the launch ended on the workgroups that own the large splats of one region (63 walk
// steps for the heaviest wave against a mean of 22 on the bench scene).  Two levels of dealing instead:
//   * a workgroup takes 8 STRIPES of 64 consecutive Gaussians that lie n_workgroups * 64 apart (stripe s of workgroup b is
//     chunk s * gridDim.x + b): every workgroup gets the same mix of regions, yet touches only ~8 small neighbourhoods of
//     tiles, so the LDS aggregation of the counters keeps most of its factor (dealing single runs over the whole grid
//     balances as well but triples the global atomics: OMFS_BIN_DEAL_GRID, measured 0.041 / 0.048 ms);
//   * inside the workgroup the 128 runs of four (64 contiguous bytes of each record array) go round-robin to the waves,
//     two runs of every stripe each, so the waves of a workgroup finish together.
static_assert(BIN_THREADS == 512, "bin_gaussian deals 8 stripes of 64 to 8 waves");
__device__ __forceinline__ int bin_gaussian(int l) {
  const int w = (int)(threadIdx.x >> 6), q = l >> 2;
#ifdef OMFS_BIN_DEAL_GRID
  const int waves = (int)gridDim.x * BIN_WAVES;
  return ((q * waves + (int)blockIdx.x * BIN_WAVES + w) << 2) + (l & 3);
#else
  const int stripe = q >> 1, run = ((q & 1) << 3) | w;          // run q * 8 + w of the workgroup's 128
#ifndef OMFS_BIN_NO_XCD_ORDER
  // which logical workgroup this one is: the grid is a multiple of 32 (bin_blocks) and workgroups go to XCD (id mod 8); logical
  // workgroup 32 a + 4 k + c runs as id 32 a + 8 c + k, i.e. on XCD k -- and every one of its stripes (chunks stripe * grid + B
  // of 64 Gaussians) was written by a project_fwd workgroup (256 Gaussians, id mod 8) of XCD k too: the records it reads
  // are still in THAT L2
  const int lin = (int)blockIdx.x, B = (lin & ~31) | ((lin & 7) << 2) | ((lin >> 3) & 3);
#else
  const int B = (int)blockIdx.x;
#endif
  return ((stripe * (int)gridDim.x + B) << 6) + (run << 2) + (l & 3);
#endif
}
// workgroups of the count / scatter launches: a multiple of 32 (see bin_gaussian; surplus workgroups own no Gaussian)
static inline int bin_blocks(int n) { return ((cdiv(n, BIN_THREADS) + 31) / 32) * 32; }

// calls f(tile) for every tile of Gaussian i's rectangle that passes the test (lane-per-Gaussian form, fallback kernels)
template <typename F>
__device__ __forceinline__ void for_each_touched_tile(const PairSource& ps, int i, int gx, F&& f) {
  const float4 r2 = ps.g2[RI(i)];
  const uint32_t rect = __float_as_uint(r2.w);
  if (rect == 0u) return;
  const float4 r0 = ps.g0[RI(i)];
  const float4 r1 = ps.g1[RI(i)];
  const int x0 = rect & 255u, y0 = (rect >> 8) & 255u, x1 = (rect >> 16) & 255u, y1 = rect >> 24;
  for (int y = y0; y < y1; ++y)
    for (int x = x0; x < x1; ++x)
      if (tile_touched(r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, x, y)) f(y * gx + x);
}

// Every lane of the wave must call this (i >= n publishes an empty rectangle).  Returns the wave's rectangle-tile total.
__device__ __forceinline__ void set_r1(float2& d, const float4& r0, const float4& t) { d = make_float2(t.x, t.y); }
__device__ __forceinline__ void set_r1(float4& d, const float4& r0, const float4& t) { d = make_float4(t.x, t.y, -r0.w / t.x, -r0.w / r0.z); }
template <bool QUOT>
__device__ __forceinline__ uint32_t publish_rects(const PairSource& ps, int i, int n, WaveRectsT<QUOT>* wr, bool* visible = nullptr) {
  const int lane = lane_id();
  uint32_t rect = 0u, depth = 0u;
  bool vis = false;
  float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f);
  typename std::conditional<QUOT, float4, float2>::type r1{};
  if (i < n) {
    const float4 r2 = ps.g2[RI(i)];
    rect = __float_as_uint(r2.w);
    depth = __float_as_uint(r2.y);
    vis = (__float_as_uint(r2.z) & 0xFFFFFu) != 0u;     // radius > 0
    if (rect) { r0 = ps.g0[RI(i)]; const float4 t = ps.g1[RI(i)]; set_r1(r1, r0, t); }
  }
  if (visible) *visible = vis;
  const uint32_t area = (((rect >> 16) & 255u) - (rect & 255u)) * ((rect >> 24) - ((rect >> 8) & 255u));
  const uint32_t incl = wave_incl_scan_u32(area, lane);
  wr->r0[lane] = r0; wr->r1[lane] = r1; wr->scan[lane] = incl; wr->rect[lane] = rect; wr->depth[lane] = depth;
  if ((lane & 7) == 7) wr->coarse[lane >> 3] = incl;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  return (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
}

// f(tile, owner lane) for every rectangle tile of the wave's 64 Gaussians that passes the frozen tile test.
// The test is evaluated once per frame: the count kernel records its outcome as one 64-bit ballot per walk step
// (HITS_RECORD, `hits` points at this wave's HITS_PER_WAVE slots in keys_tmp, which nothing else uses before the
// sort), the scatter kernel's two walks replay it (HITS_REPLAY) and skip the ~100-instruction test; waves with
// more steps than slots recompute (HITS_NONE behaviour for the steps beyond).
constexpr int HITS_PER_WAVE = 64;
enum { HITS_NONE = 0, HITS_RECORD = 1, HITS_REPLAY = 2 };

__device__ __forceinline__ bool tile_test(const float4& r0, const float2& r1, int tx, int ty) {
  return tile_touched(r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, tx, ty);
}
__device__ __forceinline__ bool tile_test(const float4& r0, const float4& r1, int tx, int ty) {
  return tile_touched_pre(r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, tx, ty);
}
template <int MODE, bool QUOT, typename F>
__device__ __forceinline__ void for_each_touched_tile_balanced(const WaveRectsT<QUOT>* wr, uint32_t total, int gx,
                                                               unsigned long long* hits, F&& f) {
  const int lane = lane_id();
  uint32_t step = 0;
  const uint4 ca = *reinterpret_cast<const uint4*>(&wr->coarse[0]), cb = *reinterpret_cast<const uint4*>(&wr->coarse[4]);
  for (uint32_t p = lane; p < total; p += 64, ++step) {
    // owner = first j with scan[j] > p = #{g : scan[g] <= p} (scan is non-decreasing).  Two 8-way levels -- the eight
    // group ends, then the eight entries of the group -- are two LDS round trips of two 16-byte reads each; the 6-step
    // bisection this replaces was a chain of six dependent LDS reads per walk step, and these kernels spend half (count)
    // to two thirds (scatter) of their wave-cycles waiting (profiles/r02_b_pmc_wait.json).
    const int jh = (int)(ca.x <= p) + (int)(ca.y <= p) + (int)(ca.z <= p) + (int)(ca.w <= p) + (int)(cb.x <= p) + (int)(cb.y <= p) +
                   (int)(cb.z <= p);                  // coarse[7] = total > p
    const uint4 fa = *reinterpret_cast<const uint4*>(&wr->scan[8 * jh]), fb = *reinterpret_cast<const uint4*>(&wr->scan[8 * jh + 4]);
    const int j = 8 * jh + (int)(fa.x <= p) + (int)(fa.y <= p) + (int)(fa.z <= p) + (int)(fa.w <= p) + (int)(fb.x <= p) +
                  (int)(fb.y <= p) + (int)(fb.z <= p);   // scan[8 jh + 7] = coarse[jh] > p
    const uint32_t rect = wr->rect[j];
    const int x0 = rect & 255u, y0 = (rect >> 8) & 255u, x1 = (rect >> 16) & 255u, y1 = rect >> 24;
    const int w = x1 - x0;
    const int k = (int)(p - (wr->scan[j] - (uint32_t)(w * (y1 - y0))));
    // k / w for k < 2^15, w < 2^8: (k + 0.5) / w is at least 0.5/w away from an integer, far above the rounding error
    const int q = (int)(((float)k + 0.5f) * __builtin_amdgcn_rcpf((float)w));
    const int tx = x0 + (k - q * w), ty = y0 + q;
    bool hit;
    if (MODE == HITS_REPLAY && hits && step < (uint32_t)HITS_PER_WAVE) {
      hit = (hits[step] >> lane) & 1ull;
    } else {
      const float4 r0 = wr->r0[j];
      hit = tile_test(r0, wr->r1[j], tx, ty);
      if (MODE == HITS_RECORD && hits && step < (uint32_t)HITS_PER_WAVE) {
        const unsigned long long bal = __ballot(hit);
        if (lane == 0) hits[step] = bal;             // lane 0 runs every step of the walk (p = 64 step)
      }
    }
    if (hit) f(ty * gx + tx, j);
  }
}

__global__ __launch_bounds__(BIN_THREADS) void bin_count_kernel(int n, PairSource ps, int gx, int n_tiles,
                                                                uint32_t* __restrict__ tile_count,
                                                                unsigned long long* __restrict__ hits_all,
                                                                uint2* __restrict__ lists_all,
                                                                uint32_t* __restrict__ n_visible, uint32_t* __restrict__ status,
                                                                uint32_t stamp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ uint32_t s_vis, s_nlist;
  if (blockIdx.x == 0 && threadIdx.x == 0) status[1] = hits_all ? stamp : 0u;   // whose tile-test ballots keys_tmp now holds
  if (threadIdx.x == 0) s_nlist = 0u;
  WaveRectsT<true>* wr = reinterpret_cast<WaveRectsT<true>*>(smem) + (threadIdx.x >> 6);
  // per-tile counters of this workgroup: 16 bits each, two per word (512 Gaussians per workgroup: a count never reaches 2^16)
  uint32_t* hist = reinterpret_cast<uint32_t*>(smem + BIN_COUNT_SCRATCH_BYTES);
  for (int t = threadIdx.x; t < (n_tiles + 1) / 2; t += BIN_THREADS) hist[t] = 0;
  if (threadIdx.x == 0) s_vis = 0u;
  const int i = bin_gaussian(lane_id());
  bool vis;
  const uint32_t total = publish_rects(ps, i, n, wr, &vis);
  __syncthreads();
  if (n_visible) {     // #Gaussians with radius > 0: one LDS atomic per wave, one global atomic per block
    const uint32_t cnt = (uint32_t)__popcll(__ballot(vis));
    if (lane_id() == 0 && cnt) atomicAdd(&s_vis, cnt);
  }
  unsigned long long* hits = hits_all ? hits_all + ((size_t)blockIdx.x * BIN_WAVES + (threadIdx.x >> 6)) * HITS_PER_WAVE : nullptr;
  for_each_touched_tile_balanced<HITS_RECORD>(wr, total, gx, hits, [&](int t, int) { atomicAdd(&hist[t >> 1], 1u << (16 * (t & 1))); });
  __syncthreads();
  // the block's non-empty (tile, count) pairs also go to a list of its own behind the ballots (entry 0: their number):
  // bin_scatter_kernel takes its slot ranges from it instead of walking every rectangle a second time just to count
  uint2* list = lists_all ? lists_all + (size_t)blockIdx.x * (size_t)(n_tiles + 1) : nullptr;
  for (int t = threadIdx.x; t < n_tiles; t += BIN_THREADS) {
    const uint32_t c = (hist[t >> 1] >> (16 * (t & 1))) & 0xFFFFu;
    if (c) {
      atomicAdd(&tile_count[t], c);
      if (list) list[1u + atomicAdd(&s_nlist, 1u)] = make_uint2((uint32_t)t, c);
    }
  }
  if (n_visible && threadIdx.x == 0 && s_vis) atomicAdd(n_visible, s_vis);
  __syncthreads();
  if (list && threadIdx.x == 0) list[0] = make_uint2(s_nlist, 0u);
}

__global__ __launch_bounds__(BIN_THREADS) void bin_scatter_kernel(int n, PairSource ps, int gx, int n_tiles,
                                                                  const uint32_t* __restrict__ tile_start,
                                                                  uint32_t* __restrict__ tile_cursor,
                                                                  uint2* __restrict__ keys,
                                                                  unsigned long long* __restrict__ hits_all,
                                                                  const uint2* __restrict__ lists_all,
                                                                  const uint32_t* __restrict__ status, uint32_t stamp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  WaveRects* wr = reinterpret_cast<WaveRects*>(smem) + (threadIdx.x >> 6);
  uint32_t* hist = reinterpret_cast<uint32_t*>(smem + BIN_SCRATCH_BYTES);  // count, then this block's next slot in the tile
  // the recorded ballots are replayed only if they are the ones omfs_bin_count left for THESE Gaussians and THIS camera
  // (stamp in status[1]; omfs_tile_sort, which reuses keys_tmp, clears it): any other call order recomputes the test
  if (status[1] != stamp) { hits_all = nullptr; lists_all = nullptr; }
  unsigned long long* hits = hits_all ? hits_all + ((size_t)blockIdx.x * BIN_WAVES + (threadIdx.x >> 6)) * HITS_PER_WAVE : nullptr;
  if (tile_start[n_tiles] == 0u) return;                // nothing visible, or capacity overflow (flagged by the scan)
  const int i = bin_gaussian(lane_id());
  if (lists_all) {
    // the count kernel left this block's (tile, count) pairs: the slot ranges come straight from them (eight returning
    // atomics in flight per thread); only the listed tiles' cursors are ever read below, so nothing is zeroed
    const uint2* list = lists_all + (size_t)blockIdx.x * (size_t)(n_tiles + 1);
    const int nl = (int)list[0].x;
    for (int e0 = threadIdx.x; e0 < nl; e0 += 8 * BIN_THREADS) {
      uint2 tc[8];
      uint32_t at[8], ts[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) { const int e = e0 + k * BIN_THREADS; tc[k] = e < nl ? list[1 + e] : make_uint2(0u, 0u); }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        at[k] = tc[k].y ? atomicAdd(&tile_cursor[tc[k].x], tc[k].y) : 0u;
        ts[k] = tc[k].y ? tile_start[tc[k].x] : 0u;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) if (tc[k].y) hist[tc[k].x] = ts[k] + at[k];
    }
  } else {
    for (int t = threadIdx.x; t < n_tiles; t += BIN_THREADS) hist[t] = 0;
  }
  const uint32_t total = publish_rects(ps, i, n, wr);
  __syncthreads();
  if (!lists_all) {
    for_each_touched_tile_balanced<HITS_REPLAY>(wr, total, gx, hits, [&](int t, int) { atomicAdd(&hist[t], 1u); });
    __syncthreads();
    // the block's slot ranges: eight returning atomics in flight per thread (one round trip per eight tiles, not one per tile)
    for (int t0 = threadIdx.x; t0 < n_tiles; t0 += 8 * BIN_THREADS) {
      uint32_t c[8], at[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) { const int t = t0 + k * BIN_THREADS; c[k] = t < n_tiles ? hist[t] : 0u; }
#pragma unroll
      for (int k = 0; k < 8; ++k) at[k] = c[k] ? atomicAdd(&tile_cursor[t0 + k * BIN_THREADS], c[k]) : 0u;
#pragma unroll
      for (int k = 0; k < 8; ++k) if (c[k]) hist[t0 + k * BIN_THREADS] = tile_start[t0 + k * BIN_THREADS] + at[k];
    }
    __syncthreads();
  }
  for_each_touched_tile_balanced<HITS_REPLAY>(wr, total, gx, hits, [&](int t, int j) {
    const uint32_t pos = atomicAdd(&hist[t], 1u);
    keys[pos] = make_uint2(wr->depth[j], (uint32_t)bin_gaussian(j));
  });
}

// Fallback for images with more tiles than fit in LDS: one global atomic per pair.
__global__ __launch_bounds__(256) void bin_count_direct_kernel(int n, PairSource ps, int gx, uint32_t* __restrict__ tile_count,
                                                               uint32_t* __restrict__ n_visible) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (n_visible) {
    const bool vis = i < n && (__float_as_uint(ps.g2[RI(i)].z) & 0xFFFFFu) != 0u;
    const uint32_t cnt = (uint32_t)__popcll(__ballot(vis));
    if (lane_id() == 0 && cnt) atomicAdd(n_visible, cnt);
  }
  if (i < n) for_each_touched_tile(ps, i, gx, [&](int t) { atomicAdd(&tile_count[t], 1u); });
}

__global__ __launch_bounds__(256) void bin_scatter_direct_kernel(int n, PairSource ps, int gx, int n_tiles,
                                                                 const uint32_t* __restrict__ tile_start,
                                                                 uint32_t* __restrict__ tile_cursor, uint2* __restrict__ keys) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || tile_start[n_tiles] == 0u) return;
  const uint32_t depth_bits = __float_as_uint(ps.g2[RI(i)].y);
  for_each_touched_tile(ps, i, gx, [&](int t) {
    keys[tile_start[t] + atomicAdd(&tile_cursor[t], 1u)] = make_uint2(depth_bits, (uint32_t)i);
  });
}

// ------------------------------------------------------------------ per-tile radix sort
// Sorts n (depth,id) pairs ascending by depth bits; ping-pongs between a and b, returns the buffer
// holding the result.  NT threads; hist: (NT/64)*256 words, misc: 8 words (both LDS).
template <int NT>
__device__ __forceinline__ uint2* radix_sort_pairs(uint2* a, uint2* b, int n, volatile uint32_t* hist,
                                                   volatile uint32_t* misc) {
  constexpr int NW = NT / 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int seg = ((n + NT - 1) / NT) * 64;  // per-wave segment, multiple of 64
  const int wbeg = min(n, wave * seg), wend = min(n, wbeg + seg);
  uint2 *src = a, *dst = b;
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = pass * 8;
    for (int k = tid; k < NW * 256; k += NT) hist[k] = 0;
    if (tid == 0) misc[0] = 0;
    __syncthreads();
    for (int k = wbeg + lane; k < wend; k += 64) atomicAdd((uint32_t*)&hist[wave * 256 + ((src[k].x >> shift) & 255u)], 1u);
    __syncthreads();
    // digit-major, wave-minor exclusive scan; thread t < 256 owns digit t
    uint32_t tot = 0, incl = 0;
    if (tid < 256) {
#pragma unroll 4
      for (int w = 0; w < NW; ++w) tot += hist[w * 256 + tid];
      incl = wave_incl_scan_u32(tot, lane);
      if (lane == 63) misc[1 + wave] = incl;
      if (tot == (uint32_t)n) misc[0] = 1;  // every key has this digit: pass is the identity
    }
    __syncthreads();
    const bool skip = misc[0] != 0;
    if (tid < 256) {
      uint32_t run = incl - tot;
      for (int w = 0; w < wave; ++w) run += misc[1 + w];
#pragma unroll 4
      for (int w = 0; w < NW; ++w) {
        const uint32_t c = hist[w * 256 + tid];
        hist[w * 256 + tid] = run;
        run += c;
      }
    }
    __syncthreads();
    if (skip) continue;
    for (int base = wbeg; base < wend; base += 64) {
      const int k = base + lane;
      const bool active = k < wend;
      uint2 item = make_uint2(0u, 0u);
      if (active) item = src[k];
      const uint32_t d = (item.x >> shift) & 255u;
      unsigned long long m = __ballot(active);
#pragma unroll
      for (int bit = 0; bit < 8; ++bit) {
        const bool set = (d >> bit) & 1u;
        const unsigned long long bal = __ballot(active && set);
        m &= set ? bal : ~bal;
      }
      const uint32_t rank = __popcll(m & ((1ull << lane) - 1ull));
      const uint32_t cnt = __popcll(m);
      uint32_t prev = 0;
      if (active) prev = hist[wave * 256 + d];
      __builtin_amdgcn_wave_barrier();
      if (active) {
        dst[prev + rank] = item;
        if (rank == 0) hist[wave * 256 + d] = prev + cnt;
      }
      __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    uint2* t = src; src = dst; dst = t;
  }
  return src;
}

// Distribution sort.  Depths inside one tile spread almost uniformly between the tile's nearest and farthest splat,
// so ONE monotone bucket pass -- bucket = floor((key - min) * NB / (max - min + 1)), NB = 4 NT buckets (about one key
// per bucket) -- followed by ranking inside each bucket, by (depth bits, id), replaces the 3-4 counting passes of a
// radix sort: the result is the same total order.  The unsorted pairs are read straight from global memory (three
// passes over 8 B per pair, L2 resident: the scatter has just written them); only the bucket-ordered copy b lives in LDS
// (or in keys_tmp for lists beyond the LDS capacity), and the ranking pass writes the ids to their final place.
// Returns false (nothing written) when some bucket holds more than BUCKET_MAX keys (many equal or clustered depths);
// the caller then falls back to the radix sort.  hist: NB words, misc: 8 words.
#ifndef OMFS_BUCKET_MAX
#define OMFS_BUCKET_MAX 256
#endif
constexpr int BUCKET_MAX = OMFS_BUCKET_MAX;

template <int NT>
__device__ __forceinline__ bool bucket_sort_to_ids(const uint2* __restrict__ src, uint2* b, int n, volatile uint32_t* vhist,
                                                   volatile uint32_t* misc, uint32_t* __restrict__ out_ids) {
  constexpr int NB = (NT / 64) * 256;
  constexpr int PER = NB / NT;   // 4 consecutive buckets per thread in the scan
  constexpr int KPT = 8;         // keys a thread keeps in registers (lists up to 8 NT pairs are read from memory ONCE)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint32_t* hist = const_cast<uint32_t*>(vhist);   // every cross-thread hand-over below goes through a barrier
  const bool in_regs = n <= KPT * NT;
  uint2 reg[KPT];
  if (in_regs) {
#pragma unroll
    for (int j = 0; j < KPT; ++j) {                // independent loads: one memory round trip
      const int k = tid + j * NT;
      reg[j] = k < n ? src[k] : make_uint2(0u, 0u);
    }
  }
  auto for_keys = [&](auto&& f) {
    if (in_regs) {
#pragma unroll
      for (int j = 0; j < KPT; ++j)
        if (tid + j * NT < n) f(reg[j]);
    } else {
      for (int k = tid; k < n; k += NT) f(src[k]);
    }
  };
  // ---- key range
  uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
  for_keys([&](const uint2& it) { kmin = min(kmin, it.x); kmax = max(kmax, it.x); });
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, d, 64));
    kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, d, 64));
  }
  for (int k = tid; k < NB; k += NT) hist[k] = 0;
  if (tid == 0) { misc[0] = 0xFFFFFFFFu; misc[1] = 0u; misc[2] = 0u; }
  __syncthreads();
  if (lane == 0) { atomicMin((uint32_t*)&misc[0], kmin); atomicMax((uint32_t*)&misc[1], kmax); }
  __syncthreads();
  kmin = misc[0];
  const float scale = (float)NB / ((float)(misc[1] - kmin) + 1.f);
  auto bucket_of = [&](uint32_t x) { return min(NB - 1, (int)((float)(x - kmin) * scale)); };   // monotone in x
  // ---- histogram
  for_keys([&](const uint2& it) { atomicAdd(&hist[bucket_of(it.x)], 1u); });
  __syncthreads();
  // ---- exclusive scan over the buckets (thread t owns buckets PER t .. PER t + PER - 1)
  uint32_t c[PER], sum = 0, big = 0;
#pragma unroll
  for (int j = 0; j < PER; ++j) { c[j] = hist[tid * PER + j]; sum += c[j]; big = max(big, c[j]); }
  const uint32_t incl = wave_incl_scan_u32(sum, lane);
  if (big > (uint32_t)BUCKET_MAX) misc[2] = 1u;
  __syncthreads();                               // every count is in registers: hist can be reused
  if (lane == 63) hist[wave] = incl;             // wave totals
  __syncthreads();
  uint32_t base = 0;
#pragma unroll
  for (int w = 0; w < NT / 64; ++w) base += w < wave ? hist[w] : 0u;
  const bool fallback = misc[2] != 0u;
  __syncthreads();                               // totals consumed before the cursors overwrite them
  if (fallback) return false;
  uint32_t run = base + incl - sum;
#pragma unroll
  for (int j = 0; j < PER; ++j) { hist[tid * PER + j] = run; run += c[j]; }   // cursor = first slot of the bucket
  __syncthreads();
  // ---- placement (order inside a bucket is arbitrary)
  for_keys([&](const uint2& it) { b[atomicAdd(&hist[bucket_of(it.x)], 1u)] = it; });
  __syncthreads();
  // ---- inside the buckets: every key counts the keys of its bucket that precede it in (depth bits, id) order
  // (all keys in parallel, reads issued four at a time) and its id goes to that rank; cursor[bkt] is now the END of bucket bkt
  for (int k = tid; k < n; k += NT) {
    const uint2 it = b[k];
    const int bkt = bucket_of(it.x);
    const int e = (int)hist[bkt], s0 = bkt ? (int)hist[bkt - 1] : 0;
    auto before = [&](const uint2& o) { return (o.x < it.x || (o.x == it.x && o.y < it.y)) ? 1 : 0; };
    int rank = 0, j = s0;
    for (; j + 4 <= e; j += 4) {
      const uint2 o0 = b[j], o1 = b[j + 1], o2 = b[j + 2], o3 = b[j + 3];
      rank += (before(o0) + before(o1)) + (before(o2) + before(o3));
    }
    for (; j < e; ++j) rank += before(b[j]);
    out_ids[s0 + rank] = it.y;
  }
  return true;
}

// Equal depth bits -> ascending Gaussian id (runs are almost always of length 1).
template <int NT>
__device__ __forceinline__ void fix_ties(uint2* s, int n) {
  for (int k = threadIdx.x; k < n; k += NT) {
    const uint32_t key = s[k].x;
    const bool start = (k == 0 || s[k - 1].x != key) && (k + 1 < n && s[k + 1].x == key);
    if (!start) continue;
    int e = k + 1;
    while (e < n && s[e].x == key) ++e;
    for (int i = k + 1; i < e; ++i) {
      const uint32_t id = s[i].y;
      int j = i - 1;
      while (j >= k && s[j].y > id) { s[j + 1].y = s[j].y; --j; }
      s[j + 1].y = id;
    }
  }
}

// grid = n_tiles (blocks walk tile_order: heavy tiles first), block = NT.  A launch handles the tiles with
// n_lo < n <= n_hi.  dynamic LDS = lds_cap*8 (the bucket-ordered pairs) + ((NT/64)*256 + 8)*4; lists longer than
// lds_cap keep that copy in keys_tmp (global) instead.
#ifdef OMFS_DEBUG_TIMELINE
// per-workgroup start / end (100 MHz real-time counter), list length and path taken: tools/sort_timeline.py
__device__ unsigned long long omfs_dbg_sort[2][3][16384];
#define OMFS_DBG_SORT_BEGIN() const unsigned long long dbg_t0 = __builtin_amdgcn_s_memrealtime()
#define OMFS_DBG_SORT_END(path)                                                                        \
  do {                                                                                                  \
    if (threadIdx.x == 0 && blockIdx.x < 16384) {                                                       \
      unsigned long long(*d)[16384] = omfs_dbg_sort[NT == 1024 ? 0 : 1];                                \
      d[0][blockIdx.x] = dbg_t0; d[1][blockIdx.x] = __builtin_amdgcn_s_memrealtime();                   \
      d[2][blockIdx.x] = (unsigned long long)n | ((unsigned long long)(path) << 32);                    \
    }                                                                                                   \
  } while (0)
#else
#define OMFS_DBG_SORT_BEGIN() do { } while (0)
#define OMFS_DBG_SORT_END(path) do { } while (0)
#endif

// One list by the first NT threads of the workgroup (the others have left): bucket sort, or the counting passes when depths cluster.
template <int NT>
__device__ __forceinline__ int sort_one_list(uint2* __restrict__ keys_s, uint2* __restrict__ tmp_s, uint32_t* __restrict__ out, int n,
                                             int lds_cap, uint2* bufB, volatile uint32_t* hist, volatile uint32_t* misc) {
  if (bucket_sort_to_ids<NT>(keys_s, n <= lds_cap ? bufB : tmp_s, n, hist, misc, out)) return n <= lds_cap ? 1 : 2;
  // clustered depths: counting passes through keys / keys_tmp (uniform over the workgroup)
  uint2* res = radix_sort_pairs<NT>(keys_s, tmp_s, n, hist, misc);
  __syncthreads();
  fix_ties<NT>(res, n);
  __syncthreads();
  for (int k = threadIdx.x; k < n; k += NT) out[k] = res[k].y;
  return 3;
}

// `small_n` (the 1024-thread instantiation): lists of up to small_n pairs are sorted by the first 512 threads -- the other eight
// waves leave at once (S_BARRIER waits for the waves of the workgroup that have not ended) -- so ONE launch serves every length
// class: the shorter lists follow the long ones in the heavy-first order and fill the slots the long lists' tail leaves empty,
// instead of waiting in a launch of their own behind a kernel boundary.  Rounds 1-2 ran two launches (1024 threads above 2048
// pairs, 512 threads and 24 KB of LDS below: 43 + 18 us); one launch with the split at 2048 / 4096 / 6144 pairs: 56 / 52 / 52 us;
// 256 threads for lists below 256 ... 2048 pairs changed nothing or lost, and launching 512-thread workgroups for everything
// (the lists beyond the LDS copy then sort with 512 threads through global memory) took 75 us.  0: every list by all NT threads.
template <int NT>
__global__ __launch_bounds__(NT) void tile_sort_kernel(const uint32_t* __restrict__ tile_order,
                                                       const uint32_t* __restrict__ tile_start,
                                                       uint2* __restrict__ keys, uint2* __restrict__ keys_tmp,
                                                       uint32_t* __restrict__ sorted_ids, int lds_cap, int n_lo, int n_hi,
                                                       uint32_t* __restrict__ status, int small_n) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NW = NT / 64;
  uint2* bufB = reinterpret_cast<uint2*>(smem);
  volatile uint32_t* hist = reinterpret_cast<volatile uint32_t*>(bufB + lds_cap);
  volatile uint32_t* misc = hist + NW * 256;
  const uint32_t tile = tile_order[blockIdx.x];
  const uint32_t s = tile_start[tile];
  const int n = (int)(tile_start[tile + 1] - s);
  if (status && blockIdx.x == 0 && threadIdx.x == 0) status[1] = 0u;   // keys_tmp is scratch from here on: no ballots to replay
  if (n <= n_lo || n > n_hi) return;
  OMFS_DBG_SORT_BEGIN();
  const int tid = threadIdx.x;
  if (n == 1) {
    if (tid == 0) sorted_ids[s] = keys[s].y;
    return;
  }
  int path;
  if (NT == 1024 && n <= small_n) {
    if (tid >= 512) return;
    path = sort_one_list<512>(keys + s, keys_tmp + s, sorted_ids + s, n, lds_cap, bufB, hist, misc);
  } else {
    path = sort_one_list<NT>(keys + s, keys_tmp + s, sorted_ids + s, n, lds_cap, bufB, hist, misc);
  }
  OMFS_DBG_SORT_END(path);
  (void)path;
}

#ifndef OMFS_SORT_SMALL_CAP
#define OMFS_SORT_SMALL_CAP 4096     // = the 8 keys per thread the 512-thread sort keeps in registers (one pass over memory)
#endif
constexpr int SORT_SMALL_CAP = OMFS_SORT_SMALL_CAP;
constexpr int SORT_LARGE_NT = 1024, SORT_LARGE_CAP_DEFAULT = 7936;   // 78 KB of LDS: two workgroups per CU
constexpr size_t sort_lds_bytes(int cap, int nt) { return (size_t)cap * 8 + ((nt / 64) * 256 + 8) * 4; }

}  // namespace omfs

using namespace omfs;

static int check_bin_args(const omfs_camera* cam, const omfs_raster_buffers* rb) {
  OMFS_REQUIRE(cam && rb, "null pointer");
  OMFS_REQUIRE(rb->g2 && rb->tile_count && rb->tile_start && rb->tile_cursor && rb->tile_order && rb->keys &&
                   rb->keys_tmp && rb->sorted_ids && rb->status, "raster buffers");
  OMFS_REQUIRE(rb->dup_capacity > 0, "dup_capacity");
  return OMFS_OK;
}

static constexpr size_t BIN_LDS_LIMIT = 150 * 1024;

// The tile-test ballots recorded by omfs_bin_count for omfs_bin_scatter live in keys_tmp (8 B per pair of capacity,
// idle until the sort): HITS_PER_WAVE words for each wave of 64 Gaussians; NULL (recompute) when it is too small.
static unsigned long long* hits_buffer(int n, const omfs_raster_buffers* rb) {
  const size_t need = (size_t)bin_blocks(n) * BIN_WAVES * HITS_PER_WAVE;
  return need <= (size_t)rb->dup_capacity ? reinterpret_cast<unsigned long long*>(rb->keys_tmp) : nullptr;
}
// ... followed by one list of (tile, count) pairs per workgroup of the count kernel (n_tiles + 1 entries each); NULL when
// keys_tmp cannot hold them (the scatter then counts for itself)
static uint2* lists_buffer(int n, int n_tiles, const omfs_raster_buffers* rb) {
  const size_t blocks = (size_t)bin_blocks(n), hits = blocks * BIN_WAVES * HITS_PER_WAVE;
  const size_t need = hits + blocks * (size_t)(n_tiles + 1);
  return need <= (size_t)rb->dup_capacity ? reinterpret_cast<uint2*>(rb->keys_tmp) + hits : nullptr;
}

// identifies (Gaussian set, camera) of a bin_count / bin_scatter pair: FNV-1a over the words both calls are given; never 0
static uint32_t hits_stamp(const omfs_gaussians* g, const omfs_camera* cam) {
  uint32_t h = 2166136261u;
  auto mix = [&](const void* p, size_t bytes) {
    const unsigned char* b = static_cast<const unsigned char*>(p);
    for (size_t i = 0; i < bytes; ++i) h = (h ^ b[i]) * 16777619u;
  };
  mix(&g->n, sizeof(g->n)); mix(&g->params, sizeof(g->params)); mix(cam, sizeof(*cam));
  return h ? h : 1u;
}

// dynamic LDS above 64 KB needs the attribute once per device and function
#include <atomic>
static int ensure_max_lds(const void* fn, int bytes, std::atomic<unsigned long long>& done) {
  int dev = 0;
  OMFS_CHECK_HIP(hipGetDevice(&dev));
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(done.load(std::memory_order_acquire) & bit)) {
    OMFS_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done.fetch_or(bit, std::memory_order_release);
  }
  return OMFS_OK;
}

extern "C" int omfs_bin_count(const omfs_gaussians* g, const omfs_camera* cam, const omfs_raster_buffers* rb, void* stream) {
  if (int rc = check_bin_args(cam, rb)) return rc;
  OMFS_REQUIRE(g && g->n > 0 && rb->g0 && rb->g1, "gaussians");
  const int gx = cdiv(cam->width, OMFS_TILE), n_tiles = gx * cdiv(cam->height, OMFS_TILE);
  hipStream_t s = (hipStream_t)stream;
  PairSource ps{(const float4*)rb->g0, (const float4*)rb->g1, (const float4*)rb->g2};
  // the LDS form is taken exactly when the scatter takes it (32-bit counters there); this kernel's own counters are 16-bit
  const size_t lds_scatter = (size_t)n_tiles * 4 + BIN_SCRATCH_BYTES, lds = (size_t)((n_tiles + 1) / 2) * 4 + BIN_COUNT_SCRATCH_BYTES;
  if (lds_scatter <= BIN_LDS_LIMIT) {
    static std::atomic<unsigned long long> attr_done{0};
    if (int rc = ensure_max_lds((const void*)bin_count_kernel, 159 * 1024, attr_done)) return rc;   // + a static word
    hipLaunchKernelGGL(bin_count_kernel, dim3(bin_blocks(g->n)), dim3(BIN_THREADS), lds, s, g->n, ps, gx, n_tiles, rb->tile_count,
                       hits_buffer(g->n, rb), lists_buffer(g->n, n_tiles, rb), rb->n_visible, rb->status, hits_stamp(g, cam));
  } else {
    hipLaunchKernelGGL(bin_count_direct_kernel, dim3(cdiv(g->n, 256)), dim3(256), 0, s, g->n, ps, gx, rb->tile_count, rb->n_visible);
  }
  OMFS_CHECK_HIP(hipGetLastError());
  return OMFS_OK;
}

extern "C" int omfs_bin_scan(const omfs_camera* cam, const omfs_raster_buffers* rb, void* stream) {
  if (int rc = check_bin_args(cam, rb)) return rc;
  const int n_tiles = cdiv(cam->width, OMFS_TILE) * cdiv(cam->height, OMFS_TILE);
  OMFS_REQUIRE(rb->order_seg0, "order_seg0");
  static std::atomic<unsigned long long> attr_done{0};
  constexpr int scan_lds_bytes = 3 * 1024 * SCAN_PER * 4;     // 96 KB
  if (int rc = ensure_max_lds((const void*)tile_scan_kernel, scan_lds_bytes, attr_done)) return rc;
  hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), scan_lds_bytes, (hipStream_t)stream, n_tiles, rb->tile_count,
                     rb->tile_start, rb->tile_cursor, rb->tile_order, rb->dup_capacity, rb->status, rb->order_seg0);
  OMFS_CHECK_HIP(hipGetLastError());
  return OMFS_OK;
}

extern "C" int omfs_bin_scatter(const omfs_gaussians* g, const omfs_camera* cam, const omfs_raster_buffers* rb,
                                void* stream) {
  if (int rc = check_bin_args(cam, rb)) return rc;
  OMFS_REQUIRE(g && g->n > 0 && rb->g0 && rb->g1, "gaussians");
  const int gx = cdiv(cam->width, OMFS_TILE), n_tiles = gx * cdiv(cam->height, OMFS_TILE);
  hipStream_t s = (hipStream_t)stream;
  PairSource ps{(const float4*)rb->g0, (const float4*)rb->g1, (const float4*)rb->g2};
  const size_t lds = (size_t)n_tiles * 4 + BIN_SCRATCH_BYTES;
  if (lds <= BIN_LDS_LIMIT) {
    static std::atomic<unsigned long long> attr_done{0};
    if (int rc = ensure_max_lds((const void*)bin_scatter_kernel, 160 * 1024, attr_done)) return rc;
    hipLaunchKernelGGL(bin_scatter_kernel, dim3(bin_blocks(g->n)), dim3(BIN_THREADS), lds, s, g->n, ps, gx, n_tiles,
                       rb->tile_start, rb->tile_cursor, (uint2*)rb->keys, hits_buffer(g->n, rb), lists_buffer(g->n, n_tiles, rb), rb->status,
                       hits_stamp(g, cam));
  } else {
    hipLaunchKernelGGL(bin_scatter_direct_kernel, dim3(cdiv(g->n, 256)), dim3(256), 0, s, g->n, ps, gx, n_tiles,
                       rb->tile_start, rb->tile_cursor, (uint2*)rb->keys);
  }
  OMFS_CHECK_HIP(hipGetLastError());
  return OMFS_OK;
}

extern "C" int omfs_tile_sort(const omfs_camera* cam, const omfs_raster_buffers* rb, void* stream) {
  if (int rc = check_bin_args(cam, rb)) return rc;
  const int n_tiles = cdiv(cam->width, OMFS_TILE) * cdiv(cam->height, OMFS_TILE);
  // 1024-thread workgroups with sort_lds_pairs pairs of LDS (default 7936 = 78 KB, two workgroups per CU); lists of up to
  // SORT_SMALL_CAP pairs use the first 512 threads only
  const int cap_large = rb->sort_lds_pairs ? (int)rb->sort_lds_pairs : SORT_LARGE_CAP_DEFAULT;
  OMFS_REQUIRE(cap_large >= 256 && sort_lds_bytes(cap_large, SORT_LARGE_NT) <= 160 * 1024, "sort_lds_pairs");
  const int cap_small = cap_large < SORT_SMALL_CAP ? cap_large : SORT_SMALL_CAP;
  static std::atomic<unsigned long long> attr_done{0};
  if (int rc = ensure_max_lds((const void*)tile_sort_kernel<SORT_LARGE_NT>, 160 * 1024, attr_done)) return rc;
  hipStream_t s = (hipStream_t)stream;
  // ONE launch of 1024-thread workgroups; lists of up to cap_small pairs are sorted by the first 512 threads of theirs
  hipLaunchKernelGGL(tile_sort_kernel<SORT_LARGE_NT>, dim3(n_tiles), dim3(SORT_LARGE_NT), sort_lds_bytes(cap_large, SORT_LARGE_NT), s,
                     rb->tile_order, rb->tile_start, (uint2*)rb->keys, (uint2*)rb->keys_tmp, rb->sorted_ids, cap_large,
                     0, 0x7fffffff, rb->status, cap_small);
  OMFS_CHECK_HIP(hipGetLastError());
  return OMFS_OK;
}

extern "C" int omfs_bin_sort(const omfs_gaussians* g, const omfs_camera* cam, const omfs_raster_buffers* rb,
                             void* stream) {
  if (int rc = omfs_bin_count(g, cam, rb, stream)) return rc;
  if (int rc = omfs_bin_scan(cam, rb, stream)) return rc;
  if (int rc = omfs_bin_scatter(g, cam, rb, stream)) return rc;
  return omfs_tile_sort(cam, rb, stream);
}

#ifdef OMFS_DEBUG_TIMELINE
extern "C" int omfs_debug_sort_timeline(int cls, unsigned long long* out, int reset) {   // out [3][16384]
  OMFS_REQUIRE(cls >= 0 && cls < 2 && out, "args");
  OMFS_CHECK_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(omfs_dbg_sort), sizeof(unsigned long long) * 3 * 16384,
                                     (size_t)cls * 3 * 16384 * sizeof(unsigned long long)));
  if (reset) {
    void* p = nullptr;
    OMFS_CHECK_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(omfs_dbg_sort)));
    OMFS_CHECK_HIP(hipMemset(p, 0, sizeof(unsigned long long) * 2 * 3 * 16384));
  }
  return OMFS_OK;
}
#endif

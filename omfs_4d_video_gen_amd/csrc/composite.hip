// Front-to-back alpha compositing, forward and backward, for gfx950 (wave64).
//
// Mapping: one 64-lane workgroup (one wave) per (16x16 tile, 8x8 quadrant); lane l owns pixel
// (l&7, l>>3) of the quadrant.  Waves are fully independent -- no workgroup barrier anywhere -- so a
// saturated quadrant retires at once and a silhouette quadrant (pixels that never saturate walk the
// whole list) holds exactly one wave slot.  Each wave streams the tile's sorted list in steps of 64
// entries through a private LDS page (gather software-pipelined one step ahead), decides per entry
// which of its four 4x4 sub-blocks the splat can reach with alpha >= 1/255 (exact ellipse-vs-band
// test with slack) and walks only those entries, with scalar bit scans over 64-bit ballots.
// Heavy tiles are dispatched first (tile_order).
//
// Spec: SURVEY.md Appendix A items 6-7: integer pixel coordinate is the sample position;
// power = -0.5(A dx^2 + C dy^2) - B dx dy, skip power > 0; alpha = min(0.99, o exp(power)), skip
// alpha < 1/255; stop a pixel before a splat that would take T below 1e-4; out = C + T bg.
// Evaluated here as alpha = exp2(power*log2e + log2 o): same value to ~1e-6 relative.
#include "composite_common.hpp"


namespace omfs {

#ifndef OMFS_FWD_SEQ_SEGS
#define OMFS_FWD_SEQ_SEGS 4
#endif
#ifndef OMFS_DEEP_WAVES
#define OMFS_DEEP_WAVES 8
#endif
constexpr int FWD_SEQ_SEGS = OMFS_FWD_SEQ_SEGS;   // list segments the one-wave forward walks before handing over
constexpr int DEEP_WAVES = OMFS_DEEP_WAVES;       // segments evaluated in parallel per deep quadrant

// Forward.  One 64-lane workgroup (= one wave) per (tile, 8x8 quadrant); lane l owns pixel (l&7, l>>3) of
// the quadrant.  No workgroup barrier exists: a wave whose pixels have all saturated simply exits and
// frees its slot, silhouette quadrants take as long as they need without holding three idle partners.
// Per step the wave gathers 64 list entries (id -> three 16-byte records; the next step's gather is in
// flight while this one is walked), converts them to the log2 domain, computes their 4-bit sub-block
// masks, publishes the records in its private LDS page and turns the mask bits into four 64-bit ballots
// held in scalar registers; it then walks ONLY the set bits (scalar bit scans) of the sub-blocks that
// still hold an unsaturated pixel.
#ifdef OMFS_FWD_WAVES
#define OMFS_FWD_ATTR __attribute__((amdgpu_waves_per_eu(OMFS_FWD_WAVES, 8)))
#else
#define OMFS_FWD_ATTR
#endif
__global__ __launch_bounds__(64) OMFS_FWD_ATTR void composite_fwd_kernel(CompCam cam, const uint32_t* __restrict__ tile_order,
                                                           const uint32_t* __restrict__ tile_start,
                                                           const uint32_t* __restrict__ sorted_ids,
                                                           const float4* __restrict__ g0, const float4* __restrict__ g1,
                                                           const float4* __restrict__ g2, float* __restrict__ image,
                                                           float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
                                                           float4* __restrict__ seg_ckpt, int keep_ckpt,
                                                           const uint32_t* __restrict__ order_seg0, uint32_t* __restrict__ seg_table,
                                                           uint32_t* __restrict__ quad_max, int depth_hint) {
  // staged records live at index 1 .. 64; index 0 is a record that no pixel can hit (log2 opacity -1e30): the visit
  // loop takes FOUR list entries per iteration and pads an incomplete batch with it (ffs of an empty bit mask is 0)
  __shared__ float4 s0[WB + 1];
  __shared__ float4 s1[WB + 1];
  __shared__ float4 s2[WB + 1];   // .x = blue (one address register serves the three reads of an entry)
  OMFS_DBG_SPAN(0);
  OMFS_DBG_PHASES();
  uint32_t upos; int quad;
  unit_quadrant_of_block(upos, quad);
  const uint32_t tile = tile_order[upos];
  const int lane = threadIdx.x;
  const int qx0 = (tile % cam.gx) * OMFS_TILE + (quad & 1) * 8, qy0 = (tile / cam.gx) * OMFS_TILE + (quad >> 1) * 8;
  const int px = qx0 + (lane & 7), py = qy0 + (lane >> 3);
  const bool inside = px < cam.width && py < cam.height;
  if (__ballot(inside) == 0ull) {
    if (quad_max && lane == 0) quad_max[tile * 4 + quad] = 0u;
    return;
  }
  const float fx = (float)px, fy = (float)py;
  float Tl = inside ? 1.f : 0.f;                 // the transmittance while the pixel takes splats, 0 once it is done (or outside the image)
  const uint32_t beg = tile_start[tile], end = tile_start[tile + 1];
#ifndef OMFS_FWD_HINT_DEPTH
#define OMFS_FWD_HINT_DEPTH 384
#endif
#ifndef OMFS_FWD_HINT_PRIO
#define OMFS_FWD_HINT_PRIO 3
#endif
  // The launch is as long as its longest walks (quadrants that never saturate: 400-512 entries), and those crawl while six
  // equally old waves share their SIMD.  Which quadrants they are is known from the last visit of this view (the caller's
  // per-view depth table: a hint, the results do not depend on it): they issue ahead of their neighbours.
  if (depth_hint && quad_max[tile * 4 + quad] >= (uint32_t)OMFS_FWD_HINT_DEPTH) __builtin_amdgcn_s_setprio(OMFS_FWD_HINT_PRIO);
  // Segment table for the backward pass (one wave per (segment, quadrant), which otherwise finds its tile by a 13-step bisection
  // of order_seg0 -- 13 dependent L2 round trips at the head of waves that visit a dozen splats): entry of global segment
  // order_seg0[upos] + k = tile | k << 16; lists of 65535 segments or more (> 8.3 M entries in one tile) get the sentinel that
  // sends the backward wave to the bisection.  Written by the tile's quadrant-0 wave, off everybody's critical path.
  if (seg_table && quad == 0) {
    const uint32_t nseg = (end - beg + OMFS_SEG - 1) / OMFS_SEG, s0g = order_seg0[upos];
    for (uint32_t k = lane; k < nseg; k += 64) seg_table[s0g + k] = nseg < 0xFFFFu ? (tile | k << 16) : 0xFFFFFFFFu;
  }
  const int sidx = ((lane >> 2) & 1) | (((lane >> 5) & 1) << 1);  // 4x4 sub-block of this lane's pixel
  unsigned long long sbl[4];
#pragma unroll
  for (int sb = 0; sb < 4; ++sb) sbl[sb] = __ballot(sidx == sb);
  float T = 1.f, C0 = 0.f, C1 = 0.f, C2 = 0.f;
  uint32_t last = 0;
  float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0;
  float r2 = 0.f;
  if (beg + lane < end) {
    const uint32_t id = sorted_ids[beg + lane];
    r0 = g0[RI(id)]; r1 = g1[RI(id)]; r2 = g2[RI(id)].x;
  }
  if (lane == 0) {
    s0[0] = make_float4(0.f, 0.f, 0.f, 0.f);
    s1[0] = make_float4(0.f, -1e30f, 0.f, 0.f);
    s2[0] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  unsigned long long live = __ballot(inside);
  // lists longer than FWD_SEQ_SEGS segments are finished by composite_fwd_deep_kernel (segment-parallel)
  const bool deep = end - beg > (uint32_t)(FWD_SEQ_SEGS * OMFS_SEG);
  const uint32_t lim = deep ? beg + (uint32_t)(FWD_SEQ_SEGS * OMFS_SEG) : end;
  for (uint32_t b = beg; b < lim && live != 0ull; b += WB) {
    // entering a new segment: checkpoint (T, C) so the backward pass can start there (see composite_bwd_kernel)
    if (keep_ckpt && b != beg && ((b - beg) & (OMFS_SEG - 1)) == 0u)
      seg_ckpt[((size_t)(beg / OMFS_SEG) + tile + (b - beg) / OMFS_SEG) * 256 + quad * 64 + lane] = make_float4(T, C0, C1, C2);
    // ---- stage this step's 64 entries
    const uint32_t k = b + lane;
    uint32_t mask = 0;
    OMFS_DBG_PHASE(3);
    OMFS_DBG_WAIT_VM();
    OMFS_DBG_PHASE(0);
    __builtin_amdgcn_wave_barrier();
    if (k < end) {
      const float A = r0.z, B = r0.w, C = r1.x;
      const float lo = __log2f(fmaxf(r1.y, 1e-30f));
      s0[lane + 1] = make_float4(r0.x, r0.y, -0.5f * LOG2E * A, -LOG2E * B);
      s1[lane + 1] = make_float4(-0.5f * LOG2E * C, lo, r1.z, r1.w);
      s2[lane + 1].x = r2;
      mask = quadrant_mask(r0.x, r0.y, A, B, C, lo, qx0, qy0);
    }
    unsigned long long ms[4];
#pragma unroll
    for (int sb = 0; sb < 4; ++sb) ms[sb] = __ballot((mask >> sb) & 1u);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (k + WB < lim) {   // next step's gather, in flight during the walk
      const uint32_t id = sorted_ids[k + WB];
      r0 = g0[RI(id)]; r1 = g1[RI(id)]; r2 = g2[RI(id)].x;
    }
    // ---- walk: splats that can touch a sub-block which still has an unsaturated pixel
    auto combine = [&](unsigned long long lv) {
      unsigned long long r = 0ull;
#pragma unroll
      for (int sb = 0; sb < 4; ++sb)
        if (lv & sbl[sb]) r |= ms[sb];
      return r;
    };
    unsigned long long m = combine(live);
    const uint32_t base = b - beg;
    OMFS_DBG_PHASE(1);
    // Four entries per iteration, no branch inside: the bit scans are scalar, the twelve LDS reads of a batch are issued
    // together, and the four alpha evaluations are independent of each other -- only the short T / stop chain is serial.
    // A wave that runs alone on its SIMD (the silhouette quadrants every launch ends on) is bound by dependent-issue
    // latency, not by throughput: the batch gives it four-fold instruction-level parallelism.  A pixel that is done, or
    // not hit, takes a no-op step (weight 0, T and last unchanged).
    while (m) {
      int jx[4];
      float4 ra[4], rc[4];
      float rb[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        jx[u] = pop_lowest_bit(m);      // 1-based staged index, 0 (the inert record) when the mask is empty
        OMFS_DBG_WORK();
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) { ra[u] = s0[jx[u]]; rc[u] = s1[jx[u]]; rb[u] = s2[jx[u]].x; }
      // Every decision below is a compare feeding a select -- no lane predicate is combined with another one.  Combined
      // predicates (hit && !stop, done || stop) compile to 64-bit scalar mask arithmetic, and the scalar unit retires
      // ONE instruction per ~4.3 cycles and SIMD however many waves are resident (tools/micro/valu_rate.hip): the walk
      // used to carry 23 scalar instructions per entry next to its 26 vector ones.
      //   the four raw alphas do not depend on the pixel state (0 unless the splat is hit) and are evaluated side by side;
      //   `Tl` is the pixel's transmittance while it takes splats and 0 once it is done (or outside the image): a done pixel
      //   "stops" again on every entry (Tn = 0), which composites nothing and leaves T, the value it ended on, alone;
      //   with alpha == 0, Tn == T and w == 0 (T >= 1e-4 always: no false stop); w > 0 exactly when the splat was composited.
      float ar[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float4 a = ra[u], c = rc[u];
        const float dx = a.x - fx, dy = a.y - fy;
        const float p2 = fma_(a.z * dx, dx, fma_(c.x * dy, dy, a.w * dx * dy));
        const float e = p2 + c.y;
        const float ev = p2 <= 0.f ? e : -1e30f;
        const float ex = fminf(0.99f, __builtin_amdgcn_exp2f(ev));
        ar[u] = ev >= LOG2_INV255 ? ex : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float alpha = ar[u];
#ifdef OMFS_DEBUG_COUNTERS
        { const unsigned long long hb = __ballot(alpha > 0.f && Tl != 0.f); OMFS_DBG_ADD(3, 1); OMFS_DBG_ADD(4, hb != 0ull); OMFS_DBG_ADD(5, __popcll(hb));
          if (jx[u]) {
            int nsb = 0, nfl = 0;
            for (int sb = 0; sb < 4; ++sb) { nsb += (hb & sbl[sb]) != 0ull; nfl += ((ms[sb] >> (jx[u] - 1)) & 1ull) && (live & sbl[sb]); }
            OMFS_DBG_ADD(16, 1); OMFS_DBG_ADD(17, nsb); OMFS_DBG_ADD(18, nfl);
            int nl = 0;
            for (int sb = 0; sb < 4; ++sb) nl += (live & sbl[sb]) != 0ull;
            if (nl) OMFS_DBG_ADD(27 + nl, 1);
          } }
#endif
        const float Tn = Tl * (1.f - alpha);
        // the splat that would take T below the threshold is not composited: weight 0, T and last stay, the pixel is done
        const bool stop = Tn < 1e-4f;
        const float w = stop ? 0.f : alpha * Tl;
        C0 = fma_(rc[u].z, w, C0);
        C1 = fma_(rc[u].w, w, C1);
        C2 = fma_(rb[u], w, C2);
        T = stop ? T : Tn;
        Tl = stop ? 0.f : Tn;
        last = w > 0.f ? base + (uint32_t)jx[u] : last;
      }
      // saturation is looked at once per batch
      const unsigned long long nl = __ballot(Tl != 0.f);
      if (nl != live) {
        live = nl;
        m &= combine(live);
      }
    }
    live = __ballot(Tl != 0.f);
    OMFS_DBG_PHASE(2);
    OMFS_DBG_STEP((b - beg) / WB);
  }
  if (deep) {
    // hand-over: the state at boundary FWD_SEQ_SEGS and, in the tile's unused boundary-0 slot, the live mask
    const size_t slot0 = (size_t)(beg / OMFS_SEG) + tile;
    if (live) seg_ckpt[(slot0 + FWD_SEQ_SEGS) * 256 + quad * 64 + lane] = make_float4(T, C0, C1, C2);
    if (lane == 0)
      seg_ckpt[slot0 * 256 + quad * 64] = make_float4(__uint_as_float((uint32_t)live), __uint_as_float((uint32_t)(live >> 32)), 0.f, 0.f);
  }
  if (inside) {
    const size_t plane = (size_t)cam.width * cam.height, o = (size_t)py * cam.width + px;
    image[o] = fma_(T, cam.bg[0], C0);
    image[plane + o] = fma_(T, cam.bg[1], C1);
    image[2 * plane + o] = fma_(T, cam.bg[2], C2);
    final_T[o] = T;
    n_contrib[o] = last;
  }
  // Depth of the quadrant = its deepest last contributor (tile-wide, 1-based): the backward wave of (segment k, this quadrant)
  // has work exactly when 128 k < depth, and then visits min(depth - 128 k, segment length) entries -- known from ONE scalar
  // load, before any of the pixel state has arrived (composite_bwd*).  A handed-over quadrant is finished by the deep kernel,
  // which raises the word.
  if (quad_max) {
    const uint32_t mx = wave_max_u32(inside ? last : 0u);
    if (lane == 0) quad_max[tile * 4 + quad] = mx;
  }
}

// Forward, deep part.  Lists longer than FWD_SEQ_SEGS segments: one workgroup of DEEP_WAVES waves per (tile,
// quadrant) that composite_fwd_kernel left unfinished.  Compositing is associative per pixel -- a segment acts on
// the incoming (T, C) as T' = T P, C' = C + T A with its own transmittance product P and colour A -- so the
// waves evaluate DEEP_WAVES consecutive segments in parallel (each from T = 1) and then every wave composes
// the partials in order (all waves hold the same pixel state in registers, so nothing is broadcast).  Only the
// stop rule is not associative: a pixel whose composed T would fall below 1e-4 inside a segment is resolved
// exactly by ONE wave in transposed form -- lanes are the segment's entries, a wave-wide prefix product of
// (1 - alpha) finds the first entry that would take T below the threshold.  Checkpoints for the backward pass
// fall out of the composition.
__device__ __forceinline__ float readlane_f(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

__device__ __forceinline__ float wave_incl_prod(float v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const float t = __shfl_up(v, d, 64);
    if (lane >= d) v *= t;
  }
  return v;
}

__global__ __launch_bounds__(DEEP_WAVES * 64) void composite_fwd_deep_kernel(
    CompCam cam, const uint32_t* __restrict__ tile_order, const uint32_t* __restrict__ tile_start,
    const uint32_t* __restrict__ sorted_ids, const float4* __restrict__ g0, const float4* __restrict__ g1,
    const float4* __restrict__ g2, float* __restrict__ image, float* __restrict__ final_T,
    uint32_t* __restrict__ n_contrib, float4* __restrict__ seg_ckpt, int keep_ckpt, uint32_t* __restrict__ quad_max) {
  __shared__ float4 pg0[DEEP_WAVES][WB + 1];  // index 0 of every wave's page: the record no pixel can hit (see composite_fwd_kernel)
  __shared__ float4 pg1[DEEP_WAVES][WB + 1];
  __shared__ float4 pg2[DEEP_WAVES][WB + 1];  // .x = blue
  __shared__ float4 comp[DEEP_WAVES][64];     // per segment and pixel: (P, A.rgb)
  __shared__ uint32_t comp_last[DEEP_WAVES][64];
  __shared__ float4 res[64];                  // exactly resolved pixels: (T, C.rgb)
  __shared__ uint32_t res_last[64];           // last contributor | terminated << 31
  OMFS_DBG_SPAN(1);
  uint32_t upos; int quad;
  unit_quadrant_of_block(upos, quad);
  const uint32_t tile = tile_order[upos];
  const uint32_t tbeg = tile_start[tile], tend = tile_start[tile + 1];
  if (tend - tbeg <= (uint32_t)(FWD_SEQ_SEGS * OMFS_SEG)) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    pg0[wave][0] = make_float4(0.f, 0.f, 0.f, 0.f);
    pg1[wave][0] = make_float4(0.f, -1e30f, 0.f, 0.f);
    pg2[wave][0] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const size_t slot0 = (size_t)(tbeg / OMFS_SEG) + tile;
  const int qx0 = (tile % cam.gx) * OMFS_TILE + (quad & 1) * 8, qy0 = (tile / cam.gx) * OMFS_TILE + (quad >> 1) * 8;
  const int px = qx0 + (lane & 7), py = qy0 + (lane >> 3);
  // A quadrant wholly outside the image was left by composite_fwd_kernel before it wrote a hand-over: its slot holds
  // whatever an earlier view left there.  Lanes outside the image are never live (the hand-over of a partly inside
  // quadrant already excludes them; the mask below makes that independent of the slot's content).
  const unsigned long long in_img = __ballot(px < cam.width && py < cam.height);
  if (in_img == 0ull) return;
  const float4 hand = seg_ckpt[slot0 * 256 + quad * 64];
  unsigned long long live = (unsigned long long)__float_as_uint(hand.x) | ((unsigned long long)__float_as_uint(hand.y) << 32);
  live = (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)live) |
         ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(live >> 32)) << 32);   // wave-uniform
  live &= in_img;
  if (live == 0ull) return;
  const float fx = (float)px, fy = (float)py;
  const size_t plane = (size_t)cam.width * cam.height, o = (size_t)py * cam.width + px;
  const int sidx = ((lane >> 2) & 1) | (((lane >> 5) & 1) << 1);
  unsigned long long sbl[4];
#pragma unroll
  for (int sb = 0; sb < 4; ++sb) sbl[sb] = __ballot(sidx == sb);
  bool done = ((live >> lane) & 1ull) == 0ull;
  const bool mine = !done;                     // pixels that stopped in the first kernel keep its output
  float T = 1.f, C0 = 0.f, C1 = 0.f, C2 = 0.f;
  uint32_t last = 0;
  if (!done) {
    const float4 ck = seg_ckpt[(slot0 + FWD_SEQ_SEGS) * 256 + quad * 64 + lane];
    T = ck.x; C0 = ck.y; C1 = ck.z; C2 = ck.w;
    last = n_contrib[o];
  }
  const uint32_t n_seg = (tend - tbeg + OMFS_SEG - 1) / OMFS_SEG;
  for (uint32_t sbatch = FWD_SEQ_SEGS; sbatch < n_seg && live != 0ull; sbatch += DEEP_WAVES) {
    const int nb = (int)min((uint32_t)DEEP_WAVES, n_seg - sbatch);
    // ---- phase A: wave w evaluates segment sbatch + w from (T, C) = (1, 0) for the pixels still live
    if (wave < nb) {
      const uint32_t sb0 = tbeg + (sbatch + wave) * OMFS_SEG, se = min(tend, sb0 + OMFS_SEG);
      float P = 1.f, A0 = 0.f, A1 = 0.f, A2 = 0.f;
      uint32_t lw = 0;
      float4 q0[2], q1[2];
      float q2[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        q0[h] = make_float4(0.f, 0.f, 0.f, 0.f); q1[h] = q0[h]; q2[h] = 0.f;
        const uint32_t k = sb0 + h * WB + lane;
        if (k < se) { const uint32_t id = sorted_ids[k]; q0[h] = g0[RI(id)]; q1[h] = g1[RI(id)]; q2[h] = g2[RI(id)].x; }
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const uint32_t b = sb0 + h * WB;
        if (b >= se) break;
        const uint32_t k = b + lane;
        uint32_t mask = 0;
        __builtin_amdgcn_wave_barrier();
        if (k < se) {
          const float A = q0[h].z, B = q0[h].w, C = q1[h].x;
          const float lo = __log2f(fmaxf(q1[h].y, 1e-30f));
          pg0[wave][lane + 1] = make_float4(q0[h].x, q0[h].y, -0.5f * LOG2E * A, -LOG2E * B);
          pg1[wave][lane + 1] = make_float4(-0.5f * LOG2E * C, lo, q1[h].z, q1[h].w);
          pg2[wave][lane + 1].x = q2[h];
          mask = quadrant_mask(q0[h].x, q0[h].y, A, B, C, lo, qx0, qy0);
        }
        unsigned long long m = 0ull;
#pragma unroll
        for (int sb = 0; sb < 4; ++sb) {
          const unsigned long long bal = __ballot((mask >> sb) & 1u);
          if (live & sbl[sb]) m |= bal;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint32_t base = b - tbeg;
        // branch-free: alpha is masked to 0 for pixels that are done or not hit (P and A then stay as they are).  One entry
        // per iteration with two alternating register sets for the LDS prefetch: the four-entry batches of
        // composite_fwd_kernel cost this kernel registers and padded slots (measured: 122 -> 144 us)
        int jn = m ? __builtin_ffsll((long long)m) : 0;
        float4 recA0 = pg0[wave][jn], recA1 = pg1[wave][jn], recB0 = recA0, recB1 = recA1;
        float recA2 = pg2[wave][jn].x, recB2 = recA2;
        auto visit = [&](const float4& a, const float4& c, const float& cb, float4& nx0, float4& nx1, float& nx2) {
          const int j = jn;
          m &= m - 1ull;
          jn = __builtin_ffsll((long long)m);          // 0 (the inert record) when nothing is left: it is not used
          nx0 = pg0[wave][jn]; nx1 = pg1[wave][jn]; nx2 = pg2[wave][jn].x;
          const float dx = a.x - fx, dy = a.y - fy;
          const float p2 = fma_(a.z * dx, dx, fma_(c.x * dy, dy, a.w * dx * dy));
          const float e = p2 + c.y;
          const bool hit = !done && p2 <= 0.f && e >= LOG2_INV255;
          const float alpha = hit ? fminf(0.99f, __builtin_amdgcn_exp2f(e)) : 0.f;
          const float w = alpha * P;
          A0 = fma_(c.z, w, A0);
          A1 = fma_(c.w, w, A1);
          A2 = fma_(cb, w, A2);
          P = P * (1.f - alpha);
          lw = hit ? base + (uint32_t)j : lw;
        };
        while (m) {
          visit(recA0, recA1, recA2, recB0, recB1, recB2);
          if (!m) break;
          visit(recB0, recB1, recB2, recA0, recA1, recA2);
        }
      }
      comp[wave][lane] = make_float4(P, A0, A1, A2);
      comp_last[wave][lane] = lw;
    }
    __syncthreads();
    // ---- phase B/C: compose in order (every wave, identically); resolve stops exactly; repeat until nothing is pending
    int nw = 0;                 // next segment of the batch this pixel has to take
    float bT = 0.f, bC0 = 0.f, bC1 = 0.f, bC2 = 0.f;
    bool have_boundary = false;
    while (true) {
      int pend = -1;
      for (int w = 0; w < nb; ++w) {
        const bool active = !done && pend < 0 && w >= nw;
        if (w == wave && active) { bT = T; bC0 = C0; bC1 = C1; bC2 = C2; have_boundary = true; }
        const float4 pc = comp[w][lane];
        const uint32_t lw = comp_last[w][lane];
        if (active) {
          const float Tn = T * pc.x;
          if (Tn < 1e-4f) {
            pend = w;                                   // stops somewhere inside this segment
          } else {
            C0 = fma_(T, pc.y, C0); C1 = fma_(T, pc.z, C1); C2 = fma_(T, pc.w, C2);
            T = Tn;
            if (lw) last = lw;
          }
        }
      }
      if (pend < 0) nw = nb;                          // took every segment of the batch (or is done)
      const unsigned long long pending = __ballot(pend >= 0);
      if (pending == 0ull) break;
      // phase C: the k-th pending pixel is resolved by wave k mod DEEP_WAVES
      unsigned long long pm = pending;
      for (int kth = 0; pm; ++kth) {
        const int pl = __builtin_ctzll(pm);
        pm &= pm - 1ull;
        if (kth % DEEP_WAVES != wave) continue;
        const int pw = __builtin_amdgcn_readlane(pend, pl);
        float Trun = readlane_f(T, pl);
        float R0 = readlane_f(C0, pl), R1 = readlane_f(C1, pl), R2 = readlane_f(C2, pl);
        uint32_t rlast = (uint32_t)__builtin_amdgcn_readlane((int)last, pl);
        const float pfx = (float)(qx0 + (pl & 7)), pfy = (float)(qy0 + (pl >> 3));
        const uint32_t sb0 = tbeg + (sbatch + (uint32_t)pw) * OMFS_SEG, se = min(tend, sb0 + OMFS_SEG);
        bool found = false;
        for (uint32_t b = sb0; b < se && !found; b += WB) {
          const uint32_t k = b + lane;
          float alpha = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f;
          bool hit = false;
          if (k < se) {
            const uint32_t id = sorted_ids[k];
            const float4 a = g0[RI(id)], c = g1[RI(id)];
            const float dx = a.x - pfx, dy = a.y - pfy;
            const float p2 = fma_((-0.5f * LOG2E * a.z) * dx, dx, fma_((-0.5f * LOG2E * c.x) * dy, dy, (-LOG2E * a.w) * dx * dy));
            const float e = p2 + __log2f(fmaxf(c.y, 1e-30f));
            if (p2 <= 0.f && e >= LOG2_INV255) {
              hit = true;
              alpha = fminf(0.99f, __builtin_amdgcn_exp2f(e));
              c0 = c.z; c1 = c.w; c2 = g2[RI(id)].x;
            }
          }
          const float om = 1.f - alpha;
          const float incl = wave_incl_prod(om, lane);
          float excl = __shfl_up(incl, 1, 64);
          if (lane == 0) excl = 1.f;
          const float Tb = Trun * excl;                   // T in front of this entry
          const bool term = hit && Tb * om < 1e-4f;
          const unsigned long long tb = __ballot(term);
          const int first = tb ? __builtin_ctzll(tb) : 64;
          const bool contrib = hit && lane < first;
          const float w = contrib ? alpha * Tb : 0.f;
          R0 += wave_sum_all(c0 * w); R1 += wave_sum_all(c1 * w); R2 += wave_sum_all(c2 * w);
          const unsigned long long cbal = __ballot(contrib);
          if (cbal) rlast = (b - tbeg) + (uint32_t)(63 - __builtin_clzll(cbal)) + 1u;
          if (tb) { found = true; Trun = readlane_f(Tb, first); }
          else Trun = Trun * readlane_f(incl, 63);
        }
        if (lane == 0) {
          res[pl] = make_float4(Trun, R0, R1, R2);
          res_last[pl] = rlast | (found ? 0x80000000u : 0u);
        }
      }
      __syncthreads();
      if (pend >= 0) {
        const float4 r = res[lane];
        const uint32_t rl = res_last[lane];
        T = r.x; C0 = r.y; C1 = r.z; C2 = r.w;
        last = rl & 0x7FFFFFFFu;
        if (rl >> 31) done = true; else nw = pend + 1;
      }
      __syncthreads();   // res is reused by the next round
    }
    // boundary checkpoints of this batch (the state in front of segment sbatch + wave)
    if (keep_ckpt && wave < nb && have_boundary)
      seg_ckpt[(slot0 + sbatch + wave) * 256 + quad * 64 + lane] = make_float4(bT, bC0, bC1, bC2);
    live = __ballot(!done);
    __syncthreads();     // comp is rewritten by the next batch
  }
  if (wave == 0 && mine) {
    image[o] = fma_(T, cam.bg[0], C0);
    image[plane + o] = fma_(T, cam.bg[1], C1);
    image[2 * plane + o] = fma_(T, cam.bg[2], C2);
    final_T[o] = T;
    n_contrib[o] = last;
  }
  if (quad_max && wave == 0) {     // the pixels finished here may lie deeper than the ones the one-wave forward finished
    const uint32_t mx = wave_max_u32(mine ? last : 0u);
    if (lane == 0) quad_max[tile * 4 + quad] = max(quad_max[tile * 4 + quad], mx);
  }
}

// Sums of 9 values over each group of 8 consecutive lanes, left in the group's last lane.  v_add_f32 with a DPP
// source (row_shr 1, 2, 4; lanes shifted in from outside read as 0) -- one instruction per value and step; the
// compiler's own lowering of the same pattern is a v_mov_dpp plus a packed add.  A VGPR written by a VALU
// instruction needs two wait states before a DPP read: the s_nop covers the first row, the 8 instructions
// between two uses of the same register cover the rest.
#ifndef OMFS_BWD_GROUP
#define OMFS_BWD_GROUP 4               // lanes per partial sum of the DPP reduction (8: three steps; 4: two steps, twice the partials)
#endif
__device__ __forceinline__ void reduce9_groups(float (&v)[9]) {
#define OMFS_DPP_ROW(SH)                                                                      \
  "v_add_f32_dpp %0, %0, %0 row_shr:" #SH " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"      \
  "v_add_f32_dpp %1, %1, %1 row_shr:" #SH " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"      \
  "v_add_f32_dpp %2, %2, %2 row_shr:" #SH " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"      \
  "v_add_f32_dpp %3, %3, %3 row_shr:" #SH " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"      \
  "v_add_f32_dpp %4, %4, %4 row_shr:" #SH " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"      \
  "v_add_f32_dpp %5, %5, %5 row_shr:" #SH " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"      \
  "v_add_f32_dpp %6, %6, %6 row_shr:" #SH " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"      \
  "v_add_f32_dpp %7, %7, %7 row_shr:" #SH " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"      \
  "v_add_f32_dpp %8, %8, %8 row_shr:" #SH " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#if OMFS_BWD_GROUP == 2
  asm volatile("s_nop 1\n\t" OMFS_DPP_ROW(1)
#elif OMFS_BWD_GROUP == 4
  asm volatile("s_nop 1\n\t" OMFS_DPP_ROW(1) OMFS_DPP_ROW(2)
#else
  asm volatile("s_nop 1\n\t" OMFS_DPP_ROW(1) OMFS_DPP_ROW(2) OMFS_DPP_ROW(4)
#endif
               : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]));
#undef OMFS_DPP_ROW
}

// Backward.  One wave per (OMFS_SEG-entry list segment, quadrant), no workgroup barrier: the serial depth
// of a silhouette quadrant is bounded by the segment length.  A pixel whose last contributor lies behind this
// segment enters it with the (T, C) the forward pass checkpointed at the next segment's start: T directly,
// and the colour behind the boundary as (C_final - C_checkpoint) / T.  The segment is walked back to front;
// per visited splat the 64 pixel contributions are reduced with two DPP steps to 16 partials per value,
// parked in one of PEND wave-private slots, and flushed 4 splats per wave-instruction: 16 lanes per 64-byte
// dsplat record, lane q < 9 sums the 16 partials of value q and adds them with one float atomic, so every
// atomic wave-instruction covers whole 64-byte segments (MI355X_MICROARCH "Global float atomics").
// Residency is what this kernel lives on (every lever that lowered it lost): 4 pending slots instead of 16 bring the
// wave-private LDS from 7.5 KB to 4 KB -- the wave slots, not the LDS, now bound the residency -- and 64 VGPRs keep all
// 8 slots per SIMD usable (0.277 -> 0.245 ms; 2, 6, 8, 12 slots: 0.251, 0.251, 0.261, 0.258).  Round 2: the reduction stops
// at 4-lane groups (18 DPP adds instead of 27; the flush sums 16 partials instead of 8) with 3 pending slots, which keeps
// the wave-private LDS at 4.5 KB = 35 waves per CU: 0.235 -> 0.230 ms (with 4 slots, 5.1 KB = 31 waves: 0.250; 2 slots: 0.241).
#ifndef OMFS_BWD_PEND
#define OMFS_BWD_PEND 3
#endif
#ifndef OMFS_BWD_WAVES
#define OMFS_BWD_WAVES 8
#endif
#define OMFS_BWD_ATTR __attribute__((amdgpu_waves_per_eu(OMFS_BWD_WAVES, 8)))
// FX: deterministic accumulation (omfs_grad_buffers.dsplat_fx).  The nine sums of a visited splat are added as 64-bit
// fixed-point integers instead of floats: integer addition is associative, so the totals do not depend on the order in which
// the waves arrive.  Everything in front of the flush is the same code.
constexpr float FX_SCALE_MOMENT = 274877906944.f;        // 2^38: columns 0..4 (moments of dL/dG G; saturate at +-2^24)
constexpr float FX_SCALE_COLOUR = 70368744177664.f;      // 2^46: columns 5..8 (d opacity, d colour; saturate at +-2^16)
__device__ __forceinline__ long long to_fixed(float v, float scale) {
  const float s = fminf(fmaxf(v * scale, -4.6e18f), 4.6e18f);     // |.| < 2^62: the cast below is defined, 2 in-range values never wrap
  return (long long)__builtin_rintf(s);
}
template <bool FX>
__global__ __launch_bounds__(64) OMFS_BWD_ATTR void composite_bwd_kernel(CompCam cam, int n_tiles, const uint32_t* __restrict__ tile_order,
                                                           const uint32_t* __restrict__ order_seg0,
                                                           const float4* __restrict__ seg_ckpt,
                                                           const uint32_t* __restrict__ tile_start,
                                                           const uint32_t* __restrict__ sorted_ids,
                                                           const float4* __restrict__ g0, const float4* __restrict__ g1,
                                                           const float4* __restrict__ g2, const float* __restrict__ image,
                                                           const float* __restrict__ final_T,
                                                           const uint32_t* __restrict__ n_contrib,
                                                           const float* __restrict__ dimage, float* __restrict__ dsplat,
                                                           const uint32_t* __restrict__ seg_table, const uint32_t* __restrict__ quad_max,
                                                           long long* __restrict__ dsplat_fx) {
  constexpr int PEND = OMFS_BWD_PEND;     // reduced splats parked before a flush
  __shared__ float4 s0[WB];
  __shared__ float4 s1[WB];
  __shared__ float2 s2[WB];               // (blue, opacity)
  __shared__ uint32_t sid[WB];
  constexpr int NGRP = 64 / OMFS_BWD_GROUP;
  __shared__ float red[PEND][NGRP][9];    // [pending slot][lane group][value]
  __shared__ uint32_t pend_id[PEND];      // Gaussian id of each pending slot
  OMFS_DBG_SPAN(2);
  uint32_t seg; int quad;
  unit_quadrant_of_block(seg, quad);
  if (seg >= order_seg0[n_tiles]) return;
  uint32_t tile, kseg;
  segment_tile(seg, n_tiles, tile_order, order_seg0, seg_table, tile, kseg);
  const int lane = threadIdx.x;
  // Depth of this quadrant (deepest last contributor of its pixels; one scalar load of the word the forward pass left): the
  // exact "does anything of this quadrant reach this segment" test and the number of entries to visit, known before any of
  // the pixel state has arrived -- so the whole head of the wave is ONE batch of loads (pixel state, checkpoint, the first
  // list entries and their records) instead of three dependent rounds through a memory system busy with gathers and atomics.
  const uint32_t qdepth = quad_max ? quad_max[tile * 4 + quad] : 0xFFFFFFFFu;
  if (qdepth <= kseg * OMFS_SEG) return;
  const uint32_t tbeg = tile_start[tile], tend = tile_start[tile + 1];
  const uint32_t beg = tbeg + kseg * OMFS_SEG, seg_len = min(tend, beg + OMFS_SEG) - beg;
  const int qx0 = (tile % cam.gx) * OMFS_TILE + (quad & 1) * 8, qy0 = (tile / cam.gx) * OMFS_TILE + (quad >> 1) * 8;
  const int px = qx0 + (lane & 7), py = qy0 + (lane >> 3);
  const bool inside = px < cam.width && py < cam.height;
  const float fx = (float)px, fy = (float)py;
  const size_t plane = (size_t)cam.width * cam.height, o = (size_t)py * cam.width + px;
  const float T_final = inside ? final_T[o] : 0.f;
  const uint32_t last_g = inside ? n_contrib[o] : 0u;     // tile-wide, 1-based
  float dL0 = 0.f, dL1 = 0.f, dL2 = 0.f, Ci0 = 0.f, Ci1 = 0.f, Ci2 = 0.f;
  if (quad_max && inside) {                               // with the depth word the wave is known to have work: load ahead
    dL0 = dimage[o]; dL1 = dimage[plane + o]; dL2 = dimage[2 * plane + o];
    Ci0 = image[o]; Ci1 = image[plane + o]; Ci2 = image[2 * plane + o];
  }
  const bool deeper = quad_max && qdepth > (kseg + 1) * OMFS_SEG;     // wave-uniform: some pixel goes on behind this segment
  float4 ck = make_float4(1.f, 0.f, 0.f, 0.f);
  if (deeper) ck = seg_ckpt[((size_t)(tbeg / OMFS_SEG) + tile + kseg + 1) * 256 + quad * 64 + lane];
  // the first step to be staged is the LAST 64-entry step of the visited range
  float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0;
  float r2 = 0.f;
  uint32_t rid = 0;
  int pre_step = -1;
  if (quad_max) {
    const uint32_t n_up = min(qdepth - kseg * OMFS_SEG, seg_len);
    pre_step = (int)((n_up - 1u) / WB);
    if ((uint32_t)lane < n_up - (uint32_t)pre_step * WB) {
      rid = sorted_ids[beg + (uint32_t)pre_step * WB + lane];
      r0 = g0[RI(rid)]; r1 = g1[RI(rid)]; r2 = g2[RI(rid)].x;
    }
  }
  if (__ballot(last_g > kseg * OMFS_SEG) == 0ull) return;   // nothing of this quadrant reaches this segment (no depth word: decided here)
  if (!quad_max && inside) {
    dL0 = dimage[o]; dL1 = dimage[plane + o]; dL2 = dimage[2 * plane + o];
    Ci0 = image[o]; Ci1 = image[plane + o]; Ci2 = image[2 * plane + o];
  }
  const uint32_t last = last_g > kseg * OMFS_SEG ? min(last_g - kseg * OMFS_SEG, seg_len) : 0u;  // segment-local
  // last contributor: maxima per 4x4 sub-block and for the quadrant bound what has to be visited
  const int sidx = ((lane >> 2) & 1) | (((lane >> 5) & 1) << 1);
  uint32_t smax[4];
#pragma unroll
  for (int sb = 0; sb < 4; ++sb) {
    uint32_t v = sidx == sb ? last : 0u;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, d, 64));
    smax[sb] = __builtin_amdgcn_readfirstlane(v);
  }
  const uint32_t n_visit = max(max(smax[0], smax[1]), max(smax[2], smax[3]));
  if (n_visit == 0) return;
  float T = T_final;
  // The colour seen behind the current splat only ever enters through its dot product with dL/dimage, so the three per-channel
  // recurrences are carried as ONE scalar per pixel, S = <colour behind, dL/dimage>.  "Behind" includes the background: behind a
  // pixel's last contributor S is <background, dL/dimage>, behind a segment boundary it is the rest of the IMAGE colour over the
  // checkpointed transmittance -- dC/dalpha_i = T_i (c_i - R_i) with R_{i-1} = alpha_i c_i + (1 - alpha_i) R_i holds with the
  // background inside R, and the separate -T_final <bg, dL> / (1 - alpha) term of the textbook form disappears.
  float S = dL0 * cam.bg[0] + dL1 * cam.bg[1] + dL2 * cam.bg[2];
  float la = 0.f, lcd = 0.f;                        // last visited splat: alpha, <colour, dL/dimage>
  if (last_g > (kseg + 1) * OMFS_SEG) {             // the pixel goes on behind this segment
    if (!deeper) ck = seg_ckpt[((size_t)(tbeg / OMFS_SEG) + tile + kseg + 1) * 256 + quad * 64 + lane];
    const float inv = __builtin_amdgcn_rcpf(ck.x);
    T = ck.x;
    S = ((Ci0 - ck.y) * dL0 + (Ci1 - ck.z) * dL1 + (Ci2 - ck.w) * dL2) * inv;
  }
  int n_pending = 0;
  auto flush_pending = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int base = 0; base < n_pending; base += 4) {
      const int slot = base + (lane >> 4), q = lane & 15;
      if (slot < n_pending && q < 9) {
        const float* src = &red[slot][0][q];
        float sum = 0.f;
#pragma unroll
        for (int p8 = 0; p8 < NGRP; ++p8) sum += src[p8 * 9];
        const float out = sum;   // moments; omfs_project_bwd turns them into d mean2d / d conic
        if (FX) {
          if (out != 0.f)
            atomicAdd(reinterpret_cast<unsigned long long*>(dsplat_fx) + (size_t)pend_id[slot] * 16 + q,
                      (unsigned long long)to_fixed(out, q < 5 ? FX_SCALE_MOMENT : FX_SCALE_COLOUR));
        } else {
          if (out != 0.f) atomicAdd(&dsplat[(size_t)pend_id[slot] * 16 + q], out);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  };
  const int n_steps = (int)((n_visit + WB - 1) / WB);
  if (n_steps - 1 != pre_step) {          // no depth word (or a stale one): gather the first step now
    const int cnt0 = (int)min((uint32_t)WB, n_visit - (uint32_t)(n_steps - 1) * WB);
    if (lane < cnt0) {
      rid = sorted_ids[beg + (uint32_t)(n_steps - 1) * WB + lane];
      r0 = g0[RI(rid)]; r1 = g1[RI(rid)]; r2 = g2[RI(rid)].x;
    }
  }
#ifdef OMFS_DEBUG_COUNTERS
  int rowtot_dbg[4] = {0, 0, 0, 0};
#endif
  for (int st = n_steps - 1; st >= 0; --st) {
    const uint32_t cbase = (uint32_t)st * WB;                       // list position of bit 0, 0-based
    const int cnt = (int)min((uint32_t)WB, n_visit - cbase);
    uint32_t mask = 0;
    __builtin_amdgcn_wave_barrier();
    if (lane < cnt) {
      const float A = r0.z, B = r0.w, C = r1.x;
      const float lo = __log2f(fmaxf(r1.y, 1e-30f));
      s0[lane] = make_float4(r0.x, r0.y, -0.5f * LOG2E * A, -LOG2E * B);
      s1[lane] = make_float4(-0.5f * LOG2E * C, lo, r1.z, r1.w);
      s2[lane] = make_float2(r2, r1.y);
      sid[lane] = rid;
      mask = quadrant_mask(r0.x, r0.y, A, B, C, lo, qx0, qy0);
    }
    // splats worth visiting: can touch a sub-block one of whose pixels has its last contributor at or
    // behind the splat (list position <= that sub-block's maximum)
    unsigned long long m = 0ull;
#ifdef OMFS_DEBUG_COUNTERS
    unsigned long long mr_dbg[4] = {0ull, 0ull, 0ull, 0ull};
#endif
#pragma unroll
    for (int sb = 0; sb < 4; ++sb) {
      const unsigned long long bal = __ballot((mask >> sb) & 1u);
      if (smax[sb] > cbase) {
        const uint32_t lim = smax[sb] - cbase;
        m |= bal & (lim >= 64u ? ~0ull : ((1ull << lim) - 1ull));
#ifdef OMFS_DEBUG_COUNTERS
        mr_dbg[sb] = bal & (lim >= 64u ? ~0ull : ((1ull << lim) - 1ull));
#endif
      }
    }
#ifdef OMFS_DEBUG_COUNTERS
    {
      const int p0 = __popcll(mr_dbg[0]), p1 = __popcll(mr_dbg[1]), p2_ = __popcll(mr_dbg[2]), p3 = __popcll(mr_dbg[3]);
      OMFS_DBG_ADD(8, __popcll(m)); OMFS_DBG_ADD(9, max(max(p0, p1), max(p2_, p3))); OMFS_DBG_ADD(10, p0 + p1 + p2_ + p3);
      rowtot_dbg[0] += p0; rowtot_dbg[1] += p1; rowtot_dbg[2] += p2_; rowtot_dbg[3] += p3;
    }
#endif
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (st > 0) {   // every earlier step is full
      rid = sorted_ids[beg + cbase - WB + lane];
      r0 = g0[RI(rid)]; r1 = g1[RI(rid)]; r2 = g2[RI(rid)].x;
    }
    // The record of the next splat is fetched from LDS while the current one is evaluated; two register sets
    // alternate (the loop body is written once and instantiated twice), so no copies are needed.
    int jbn = m ? 63 - __builtin_clzll(m) : 0;
    float4 recA0 = s0[jbn], recA1 = s1[jbn], recB0 = recA0, recB1 = recA1;
    float2 recA2 = s2[jbn], recB2 = recA2;
    auto visit = [&](const float4& an, const float4& cn, const float2& cbn, float4& nx0, float4& nx1, float2& nx2) {
      const int jb = jbn;
      OMFS_DBG_WORK();
      m &= ~(1ull << jb);
      const uint32_t contributor = cbase + (uint32_t)jb + 1u;  // 1-based list position
      float v[9];
      const float4 a = an;
      const float4 c = cn;
      const float2 cb = cbn;
      jbn = 63 - __builtin_clzll(m | 1ull);   // prefetch the next splat's record
      nx0 = s0[jbn]; nx1 = s1[jbn]; nx2 = s2[jbn];
      const float dx = a.x - fx, dy = a.y - fy;
      const float p2 = fma_(a.z * dx, dx, fma_(c.x * dy, dy, a.w * dx * dy));
      const float e = p2 + c.y;
      const bool hit = contributor <= last && p2 <= 0.f && e >= LOG2_INV255;
      const unsigned long long hit_bal = __ballot(hit);
      OMFS_DBG_ADD(0, 1); OMFS_DBG_ADD(1, hit_bal != 0ull); OMFS_DBG_ADD(2, __popcll(hit_bal));
#ifdef OMFS_DEBUG_COUNTERS
      {
        const unsigned long long geo = __ballot(p2 <= 0.f && e >= LOG2_INV255);
        OMFS_DBG_ADD(12, __popcll(geo));
        int nsb = 0;
        for (int sb = 0; sb < 4; ++sb) nsb += (hit_bal & __ballot(sidx == sb)) != 0ull;
        OMFS_DBG_ADD(13, nsb);
        int na = 0;                    // sub-blocks whose deepest last contributor lies at or behind this entry
        for (int sb = 0; sb < 4; ++sb) na += smax[sb] >= contributor;
        OMFS_DBG_ADD(19 + na, 1); OMFS_DBG_ADD(23 + na, __popcll(hit_bal));
      }
#endif
      if (hit_bal == 0ull) return;    // nobody in this quadrant was touched: nothing to reduce
      {
        // Branch-free: G is masked to 0 for lanes that are not hit, which makes alpha = 0, 1/(1-alpha) = 1 and every
        // gradient term below exactly 0 for them; their colour recurrence takes a no-op step (a splat of alpha 0).
        const float G = hit ? __builtin_amdgcn_exp2f(p2) : 0.f;
        const float oG = cb.y * G;                              // opacity * G
        const float alpha = fminf(0.99f, oG);
        const float r1a = __builtin_amdgcn_rcpf(1.f - alpha);   // 1/(1-alpha), ~1 ulp
        T = T * r1a;
        const float w = alpha * T;
        v[6] = w * dL0; v[7] = w * dL1; v[8] = w * dL2;  // dL/dcolour
        S = fma_(la, lcd - S, S);                               // the splat behind this one joins what is seen behind
        const float cd = fma_(cb.x, dL2, fma_(c.w, dL1, c.z * dL0));
        lcd = cd; la = alpha;
        const float dLa = (cd - S) * T;
        // alpha = min(0.99, o*G) is differentiated straight through the clamp, as the upstream rasteriser does
        // (DESIGN.md "Frozen conventions").  Only the moments of gL = dL/dG * G over the pixels are reduced here:
        //   S_x = sum gL dx, S_y, S_xx, S_xy, S_yy  (dx = mean - pixel);
        // omfs_project_bwd turns them into d mean2d = -(A S_x + B S_y, C S_y + B S_x) and
        // d conic = (-S_xx / 2, -S_xy, -S_yy / 2), once per Gaussian.
        const float gL = oG * dLa;                     // opacity folded in
        v[0] = gL * dx;
        v[1] = gL * dy;
        v[2] = v[0] * dx;
        v[3] = v[0] * dy;
        v[4] = v[1] * dy;
        v[5] = G * dLa;                                 // d opacity
      }
      reduce9_groups(v);   // the last lane of every OMFS_BWD_GROUP-lane group holds the group's sums
      if ((lane & (OMFS_BWD_GROUP - 1)) == OMFS_BWD_GROUP - 1) {
        float* dst = &red[n_pending][lane / OMFS_BWD_GROUP][0];
#pragma unroll
        for (int q = 0; q < 9; ++q) dst[q] = v[q];
      }
      if (lane == 0) pend_id[n_pending] = sid[jb];
      if (++n_pending == PEND) { flush_pending(); n_pending = 0; }
    };
    while (m) {
      visit(recA0, recA1, recA2, recB0, recB1, recB2);
      if (!m) break;
      visit(recB0, recB1, recB2, recA0, recA1, recA2);
    }
  }
  if (n_pending) flush_pending();
#ifdef OMFS_DEBUG_COUNTERS
  OMFS_DBG_ADD(11, max(max(rowtot_dbg[0], rowtot_dbg[1]), max(rowtot_dbg[2], rowtot_dbg[3])));
  OMFS_DBG_ADD(14, 1);
#endif
}

// Deterministic mode: the fixed-point totals become the float records omfs_project_bwd reads; the accumulator is left zero.
__global__ void dsplat_from_fixed_kernel(long long* __restrict__ fx, float* __restrict__ dsplat, size_t n16) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n16) return;
  const int q = (int)(i & 15);
  if (q >= 9) return;
  const long long v = fx[i];
  if (v == 0) return;
  fx[i] = 0;
  dsplat[i] = (float)((double)v * (q < 5 ? 1.0 / 274877906944.0 : 1.0 / 70368744177664.0));
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

// Training targets kept as 8-bit RGB in HBM (a quarter of the fp32 size: thousands of 1080p views stay resident) are
// expanded to the planar fp32 layout of the loss kernels one view at a time: value / 255, exactly what a host-side
// conversion gives.
__global__ void rgb8_to_image_kernel(const uint8_t* __restrict__ rgb8, int width, int height, float* __restrict__ image) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = width * height;
  if (i >= n) return;
#pragma unroll
  for (int c = 0; c < 3; ++c) image[(size_t)c * n + i] = (float)rgb8[(size_t)i * 3 + c] / 255.0f;
}

// The same 8-bit conversion laid out as the raw scanlines of a PNG: every row starts with filter byte 0, so the host only
// has to deflate the buffer (no per-frame reshuffle under the interpreter lock).
__global__ void image_to_png_rows_kernel(const float* __restrict__ image, int width, int height, uint8_t* __restrict__ rows) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = width * height;
  if (i >= n) return;
  const int y = i / width, x = i - y * width;
  uint8_t* o = rows + (size_t)y * (1 + 3 * (size_t)width) + 1 + 3 * (size_t)x;
  if (x == 0) o[-1] = 0;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float v = fminf(fmaxf(image[(size_t)c * n + i], 0.f), 1.f);
    o[c] = (uint8_t)(v * 255.f + 0.5f);
  }
}

}  // namespace omfs

using namespace omfs;

extern "C" int omfs_composite_fwd(const omfs_camera* cam, const omfs_raster_buffers* rb, void* stream) {
  OMFS_REQUIRE(cam && rb, "null pointer");
  OMFS_REQUIRE(rb->g0 && rb->g1 && rb->g2 && rb->tile_order && rb->tile_start && rb->sorted_ids && rb->image &&
                   rb->final_T && rb->n_contrib && rb->seg_ckpt, "raster buffers");
  CompCam cc = make_compcam(cam);
  const int n_tiles = cc.gx * cdiv(cam->height, OMFS_TILE);
  OMFS_REQUIRE(rb->seg_capacity >= (uint32_t)n_tiles + rb->dup_capacity / OMFS_SEG, "seg_capacity < n_tiles + dup_capacity/OMFS_SEG");
  const int keep_ckpt = (rb->flags & OMFS_RB_FORWARD_ONLY) ? 0 : 1;
  hipLaunchKernelGGL(composite_fwd_kernel, dim3(n_tiles * 4), dim3(64), 0, (hipStream_t)stream, cc, rb->tile_order,
                     rb->tile_start, rb->sorted_ids, (const float4*)rb->g0, (const float4*)rb->g1,
                     (const float4*)rb->g2, rb->image, rb->final_T, rb->n_contrib, (float4*)rb->seg_ckpt, keep_ckpt,
                     rb->order_seg0, keep_ckpt ? segment_table(rb) : nullptr, keep_ckpt ? quadrant_depths(rb, n_tiles) : nullptr,
                     (keep_ckpt && rb->quad_depth && !(rb->flags & OMFS_RB_NO_DEPTH_HINT)) ? 1 : 0);
  OMFS_CHECK_HIP(hipGetLastError());
  // quadrants whose list is longer than FWD_SEQ_SEGS segments and still unsaturated (the rest exit at once)
  hipLaunchKernelGGL(composite_fwd_deep_kernel, dim3(n_tiles * 4), dim3(DEEP_WAVES * 64), 0, (hipStream_t)stream, cc, rb->tile_order,
                     rb->tile_start, rb->sorted_ids, (const float4*)rb->g0, (const float4*)rb->g1,
                     (const float4*)rb->g2, rb->image, rb->final_T, rb->n_contrib, (float4*)rb->seg_ckpt, keep_ckpt,
                     keep_ckpt ? quadrant_depths(rb, n_tiles) : nullptr);
  OMFS_CHECK_HIP(hipGetLastError());
  return OMFS_OK;
}

extern "C" int omfs_composite_bwd(const omfs_camera* cam, const omfs_raster_buffers* rb, const omfs_grad_buffers* gb,
                                  void* stream) {
  OMFS_REQUIRE(cam && rb && gb, "null pointer");
  OMFS_REQUIRE(rb->g0 && rb->g1 && rb->g2 && rb->tile_order && rb->tile_start && rb->sorted_ids && rb->image &&
                   rb->final_T && rb->n_contrib && rb->seg_ckpt && gb->dimage && gb->dsplat, "buffers");
  CompCam cc = make_compcam(cam);
  const int n_tiles = cc.gx * cdiv(cam->height, OMFS_TILE);
  OMFS_REQUIRE(rb->order_seg0 && rb->seg_capacity >= (uint32_t)n_tiles + rb->dup_capacity / OMFS_SEG, "segment buffers");
  // one wave per (list segment, quadrant); the grid covers the segment capacity, waves beyond the device-side total
  // (order_seg0[n_tiles]) exit at once.  ONE implementation lives in this library; the second opinions the tests hold it
  // against (matrix-core reduction, lanes = list entries) are built into libomfs_experiments.so (composite_experiments.hip).
  const dim3 grid(rb->seg_capacity * 4);
  if (!gb->dsplat_fx) {
    hipLaunchKernelGGL(composite_bwd_kernel<false>, grid, dim3(64), 0, (hipStream_t)stream, cc, n_tiles,
                       rb->tile_order, rb->order_seg0, (const float4*)rb->seg_ckpt, rb->tile_start, rb->sorted_ids, (const float4*)rb->g0,
                       (const float4*)rb->g1, (const float4*)rb->g2, rb->image, rb->final_T, rb->n_contrib, gb->dimage, gb->dsplat,
                       segment_table(rb), quadrant_depths(rb, n_tiles), (long long*)nullptr);
  } else {
    // deterministic accumulation: fixed-point integer atomics, then the totals into the float records
    OMFS_REQUIRE(gb->n_records > 0, "deterministic accumulation needs omfs_grad_buffers.n_records (Gaussians the records were projected for)");
    hipLaunchKernelGGL(composite_bwd_kernel<true>, grid, dim3(64), 0, (hipStream_t)stream, cc, n_tiles,
                       rb->tile_order, rb->order_seg0, (const float4*)rb->seg_ckpt, rb->tile_start, rb->sorted_ids, (const float4*)rb->g0,
                       (const float4*)rb->g1, (const float4*)rb->g2, rb->image, rb->final_T, rb->n_contrib, gb->dimage, gb->dsplat,
                       segment_table(rb), quadrant_depths(rb, n_tiles), gb->dsplat_fx);
    OMFS_CHECK_HIP(hipGetLastError());
    const size_t n16 = (size_t)gb->n_records * 16;
    hipLaunchKernelGGL(dsplat_from_fixed_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, gb->dsplat_fx, gb->dsplat, n16);
  }
  OMFS_CHECK_HIP(hipGetLastError());
  return OMFS_OK;
}

#ifdef OMFS_DEBUG_TIMELINE
extern "C" int omfs_debug_timeline(int kernel, unsigned long long* out, int n, int reset) { return dbg_timeline_read(kernel, out, n, reset); }
#endif
#ifdef OMFS_DEBUG_COUNTERS
extern "C" int omfs_debug_counters(unsigned long long* out8, int reset) { return dbg_counters_read(out8, reset); }
#endif

extern "C" int omfs_image_to_rgb8(const float* image, int width, int height, uint8_t* rgb8, void* stream) {
  OMFS_REQUIRE(image && rgb8 && width > 0 && height > 0, "args");
  hipLaunchKernelGGL(image_to_rgb8_kernel, dim3(cdiv(width * height, 256)), dim3(256), 0, (hipStream_t)stream, image,
                     width, height, rgb8);
  OMFS_CHECK_HIP(hipGetLastError());
  return OMFS_OK;
}

extern "C" int omfs_image_to_png_rows(const float* image, int width, int height, uint8_t* rows, void* stream) {
  OMFS_REQUIRE(image && rows && width > 0 && height > 0, "args");
  hipLaunchKernelGGL(image_to_png_rows_kernel, dim3(cdiv(width * height, 256)), dim3(256), 0, (hipStream_t)stream, image,
                     width, height, rows);
  OMFS_CHECK_HIP(hipGetLastError());
  return OMFS_OK;
}

extern "C" int omfs_rgb8_to_image(const uint8_t* rgb8, int width, int height, float* image, void* stream) {
  OMFS_REQUIRE(rgb8 && image && width > 0 && height > 0, "args");
  hipLaunchKernelGGL(rgb8_to_image_kernel, dim3(cdiv(width * height, 256)), dim3(256), 0, (hipStream_t)stream, rgb8, width,
                     height, image);
  OMFS_CHECK_HIP(hipGetLastError());
  return OMFS_OK;
}

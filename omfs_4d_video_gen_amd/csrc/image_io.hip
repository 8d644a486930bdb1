// Image ingress for the engine: a decoded 8-bit training image (RGB or RGBA, optionally a separate matte) becomes the
// target the loss kernels read -- resized to the training resolution, composited on the run's background, laid out as
// planar fp32 or interleaved 8-bit -- in one pass on the device.  The same routine produces the `gt/` images render.py
// writes next to its renders, so an evaluator compares a render with exactly what the trainer was shown
// (`02_Visual_Engine/validation_reporting.py:60-78`; upstream's loader resizes with PIL and composites `image * alpha + bg *
// (1 - alpha)`, the call site being `train_ghost.py:227-240`: --resolution and --white_background are passed through).
//
// Resize rule = PIL's Image.resize(..., Image.BOX): along an axis of scale s = in / out, output x averages, with equal
// weights, the source pixels x' in [int(x s + 0.5), int((x + 1) s + 0.5)) whose centre lies inside the footprint
// (-0.5 < (x' + 0.5 - (x + 0.5) s) / s <= 0.5).  PIL rounds to 8 bits after each of its two passes, this kernel once: the
// results differ by at most one level (tests/test_gpu_targets.py).  Without a resize every step is the identity.
#include "common.hpp"

namespace omfs {

struct AxisBox { int lo, n; };

__device__ __forceinline__ AxisBox box_range(int x, double scale, int in_size) {
  const double support = 0.5 * (scale > 1.0 ? scale : 1.0), center = (x + 0.5) * scale;
  int lo = (int)(center - support + 0.5), hi = (int)(center + support + 0.5);
  lo = lo < 0 ? 0 : lo;
  hi = hi > in_size ? in_size : hi;
  return AxisBox{lo, hi - lo};
}
__device__ __forceinline__ float box_weight(int xs, int x, double scale) {
  const double fs = scale > 1.0 ? scale : 1.0;
  const double t = ((double)xs + 0.5 - ((double)x + 0.5) * scale) / fs;
  return (t > -0.5 && t <= 0.5) ? 1.f : 0.f;
}

// one thread per output pixel
__global__ void prepare_target_kernel(const uint8_t* __restrict__ src, int channels, int sw, int sh, const uint8_t* __restrict__ mask,
                                      int dw, int dh, float bg0, float bg1, float bg2, float* __restrict__ out_f32,
                                      uint8_t* __restrict__ out_u8) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= dw * dh) return;
  const int y = i / dw, x = i - y * dw;
  const double sx = (double)sw / dw, sy = (double)sh / dh;
  const AxisBox bx = box_range(x, sx, sw), by = box_range(y, sy, sh);
  float acc[4] = {0.f, 0.f, 0.f, 0.f}, macc = 0.f, wsum = 0.f;
  for (int yy = 0; yy < by.n; ++yy) {
    const float wy = box_weight(by.lo + yy, y, sy);
    if (wy == 0.f) continue;
    for (int xx = 0; xx < bx.n; ++xx) {
      const float w = wy * box_weight(bx.lo + xx, x, sx);
      if (w == 0.f) continue;
      const size_t o = (size_t)(by.lo + yy) * sw + (bx.lo + xx);
      for (int c = 0; c < channels; ++c) acc[c] += w * (float)src[o * channels + c];
      if (mask) macc += w * (float)mask[o];
      wsum += w;
    }
  }
  const float inv = wsum > 0.f ? 1.f / wsum : 0.f;
  float rgb[3], m = 1.f;
  for (int c = 0; c < 3; ++c) rgb[c] = rintf(acc[c < channels ? c : channels - 1] * inv);     // 8-bit levels, as a resized PNG holds
  bool matte = false;
  if (mask) { m = rintf(macc * inv) * (1.f / 255.f); matte = true; }
  else if (channels == 4) { m = rintf(acc[3] * inv) * (1.f / 255.f); matte = true; }
  const float bg[3] = {bg0, bg1, bg2};
  const size_t n = (size_t)dw * dh;
  for (int c = 0; c < 3; ++c) {
    if (out_f32) {
      const float v = rgb[c] * (1.f / 255.f);
      out_f32[(size_t)c * n + i] = matte ? v * m + (1.f - m) * bg[c] : v;
    }
    if (out_u8) {
      const float v = matte ? rintf(rgb[c] * m + (1.f - m) * (255.f * bg[c])) : rgb[c];
      out_u8[(size_t)i * 3 + c] = (uint8_t)fminf(fmaxf(v, 0.f), 255.f);
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// PNG egress on the device (render_surgery is encode-bound once the frames exist,
// `02_Visual_Engine/render_surgery.py:324-362` reads them back as PNG files).  The scanlines omfs_image_to_png_rows lays
// out ([H][1 + 3W] bytes, filter type 0) are turned into a complete zlib stream here; the host adds the PNG chunk framing and
// the chunk CRC and writes the file.
//
// Deflate with the FIXED Huffman code and run-length matches at distance 3 (one RGB pixel back: a constant background of
// any colour is a run; the Z_RLE strategy the host encoder used only sees distance 1).  One 256-thread workgroup per scanline:
//   1. the row is staged in LDS with coalesced word loads; thread t owns piece t of it (row_bytes / 256 bytes, a multiple of 4);
//   2. pass 1 walks the piece greedily (run of >= 4 bytes equal to the byte 3 back -> one length/distance pair, else a
//      literal) and counts bits; a wave-wide prefix sum gives every lane its bit offset;
//   3. pass 2 walks again and ORs its codes into the LDS image of the row's block (LDS atomics: neighbouring lanes share words);
//   4. the block is closed with the end-of-block code and an EMPTY STORED BLOCK, which pads to a byte boundary (the zlib
//      "sync flush" marker), so the rows' blocks can be concatenated bytewise whatever their bit lengths;
//   5. the row's Adler-32 terms (sum of bytes, position-weighted sum) are reduced for the stream's checksum.
// omfs_png_assemble then places row r at 2 + sum(sizes[< r]) (every wave sums the sizes in front of it: no scan launch), and one
// extra wave writes the zlib header, the final empty block, the Adler-32 of all scanlines and the stream length.
constexpr uint32_t ADLER_MOD = 65521u;

__device__ __forceinline__ uint32_t bitrev(uint32_t v, int n) { return __brev(v) >> (32 - n); }

// fixed-Huffman code of a literal / length symbol, already bit-reversed for the LSB-first stream: (bits, count)
__device__ __forceinline__ void lit_code(uint32_t sym, uint32_t& bits, int& n) {
  if (sym < 144u) { bits = bitrev(0x30u + sym, 8); n = 8; }
  else if (sym < 256u) { bits = bitrev(0x190u + (sym - 144u), 9); n = 9; }
  else if (sym < 280u) { bits = bitrev(sym - 256u, 7); n = 7; }
  else { bits = bitrev(0xC0u + (sym - 280u), 8); n = 8; }
}
// length 3..258 -> its length symbol + extra bits, then the distance symbol of distance 3 (code 2, five bits, no extra bits)
__device__ __forceinline__ void match_code(int len, unsigned long long& bits, int& n) {
  uint32_t sym, extra = 0;
  int ebits = 0;
  if (len == 258) sym = 285u;
  else if (len <= 10) sym = 254u + (uint32_t)len;
  else {
    const int l3 = len - 3;
    ebits = (31 - __clz(l3)) - 2;                       // 11..18 -> 1, 19..34 -> 2, ... 131..257 -> 5
    sym = 257u + (uint32_t)(4 * ebits) + (uint32_t)((l3 >> ebits) & 3) + 4u;
    extra = (uint32_t)l3 & ((1u << ebits) - 1u);
  }
  uint32_t cb; int cn;
  lit_code(sym, cb, cn);
  bits = (unsigned long long)cb | ((unsigned long long)extra << cn) | ((unsigned long long)bitrev(2u, 5) << (cn + ebits));
  n = cn + ebits + 5;
}

// ---- tokenisation of a thread's piece without data-dependent inner loops.  A byte-by-byte walk (compare with the byte three
// back, extend the run, else emit a literal) is a chain of dependent LDS byte loads inside doubly divergent loops: 171 us per
// 1080p frame with one wave per scanline.  Instead every piece (<= 64 bytes) is first turned into two 64-bit masks with
// independent aligned word loads -- eq: byte equals the byte 3 back, ge: byte >= 144 (its literal code has 9 bits) -- and the
// walk is bit arithmetic: the next position where a run of >= 4 starts is a count-trailing-zeros of eq & eq>>1 & eq>>2 & eq>>3,
// the literals in front of it cost 8 bits each plus a popcount of ge, the run length is a count-trailing-zeros of ~eq.
// FOUR waves share a scanline (pieces of 24 bytes at 1080p: the serial work of a wave, which is what the kernel's time is,
// shrinks accordingly), and pieces that lie wholly inside a run are merged along the wave: the first thread of such a chain
// emits the whole stretch as 258-byte matches, the others nothing -- a plain background costs ~40 bytes per row.
__device__ __forceinline__ uint32_t movemask4(uint32_t y) {  // bits 7, 15, 23, 31 -> bits 0..3
  const uint32_t z = y >> 7;
  return (z | (z >> 7) | (z >> 14) | (z >> 21)) & 0xFu;
}
__device__ __forceinline__ int ctz64(unsigned long long v) { return v ? __builtin_ctzll(v) : 64; }

struct PieceMasks { unsigned long long eq, ge, eq4, valid; int len; };
// row32: the row's bytes as aligned words in LDS; [sb, sb + len) the piece (sb a multiple of 4, len <= 64)
__device__ __forceinline__ PieceMasks build_masks(const uint32_t* row32, int sb, int len) {
  PieceMasks m;
  m.len = len;
  unsigned long long eq = 0ull, ge = 0ull;
  const int base = sb >> 2, ndw = (len + 3) >> 2;
#pragma unroll
  for (int d = 0; d < 16; ++d) {
    if (d < ndw) {
      const uint32_t cur = row32[base + d], prev = base + d > 0 ? row32[base + d - 1] : 0u;
      const uint32_t x = cur ^ ((prev >> 8) | (cur << 24));                      // byte k against byte k - 3
      const uint32_t zero = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);   // 0x80 in every zero byte
      const uint32_t big = ((cur & 0x7F7F7F7Fu) + 0x70707070u) & cur & 0x80808080u;   // 0x80 in every byte >= 144
      eq |= (unsigned long long)movemask4(zero) << (4 * d);
      ge |= (unsigned long long)movemask4(big) << (4 * d);
    }
  }
  if (sb == 0) eq &= ~7ull;                                    // the first three bytes of a row have no byte three back
  m.valid = len >= 64 ? ~0ull : ((1ull << len) - 1ull);
  m.eq = eq & m.valid;
  m.ge = ge & m.valid;
  m.eq4 = m.eq & (m.eq >> 1) & (m.eq >> 2) & (m.eq >> 3);
  return m;
}

// walk a piece: lits(i, n) for every maximal stretch of literals [i, i + n), match(len) for every run
template <typename Lits, typename Match>
__device__ __forceinline__ void walk_piece(const PieceMasks& m, Lits&& lits, Match&& match) {
  int i = 0;
  while (i < m.len) {
    const int nlit = min(ctz64(m.eq4 >> i), m.len - i);
    if (nlit) { lits(i, nlit); i += nlit; }
    if (i < m.len) {
      const int run = min(ctz64(~(m.eq >> i)), m.len - i);     // >= 4: eq4 has bit i set; <= 64 < 258
      match(run);
      i += run;
    }
  }
}
// a stretch of `total` >= 4 bytes inside a run, as matches of <= 258 bytes none of which is shorter than 3
template <typename Match>
__device__ __forceinline__ void emit_long_run(int total, Match&& match) {
  while (total > 0) {
    int r = min(total, 258);
    const int rem = total - r;
    if (rem > 0 && rem < 3) r -= 3 - rem;
    match(r);
    total -= r;
  }
}

constexpr int PNG_NT = 256;
__global__ __launch_bounds__(PNG_NT) void png_deflate_rows_kernel(const uint8_t* __restrict__ rows, int row_bytes, int height,
                                                                  uint8_t* __restrict__ slots, int slot_stride,
                                                                  uint32_t* __restrict__ sizes, uint32_t* __restrict__ adler) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  __shared__ uint32_t s_wave[PNG_NT / 64], s_ad[PNG_NT / 64][2], s_misc[4];      // 64 bytes: the dynamic region stays 16-byte aligned
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row_pad = (row_bytes + 8 + 15) / 16 * 16;
  uint32_t* row32 = reinterpret_cast<uint32_t*>(lds);               // the row, byte 0 at word 0
  const uint8_t* row8 = lds;
  uint32_t* s_out = reinterpret_cast<uint32_t*>(lds + row_pad);     // [slot_stride / 4] words of the row's deflate block
  const uint8_t* src = rows + (size_t)r * row_bytes;
  // stage: a row starts at an arbitrary byte offset of the scanline buffer -> aligned word loads, funnel-shifted into place
  const int mis = (int)(reinterpret_cast<uintptr_t>(src) & 3u);
  const uint32_t* src32 = reinterpret_cast<const uint32_t*>(src - mis);
  const int n_words = (row_bytes + 3) / 4;
  for (int i0 = 0; i0 <= n_words; i0 += PNG_NT * 4) {               // four words per thread in flight
    uint32_t lo[4], hi[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * PNG_NT + tid;
      lo[u] = i < n_words ? src32[i] : 0u;
      hi[u] = (i < n_words && mis) ? src32[i + 1] : 0u;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * PNG_NT + tid;
      if (i > n_words) continue;                                    // word n_words: padding behind the row (zero)
      uint32_t v = mis ? __funnelshift_r(lo[u], hi[u], 8 * mis) : lo[u];
      const int tail = row_bytes - 4 * i;                           // bytes of this word that belong to the row
      if (tail < 4) v &= tail <= 0 ? 0u : ((1u << (8 * tail)) - 1u);
      row32[i] = v;
    }
  }
  for (int i = tid; i < slot_stride / 4; i += PNG_NT) s_out[i] = 0u;
  __syncthreads();
  // pieces: a multiple of 4 bytes, at least 16 (short rows use fewer threads), at most 64 (one mask word)
  const int piece = max(16, ((row_bytes + PNG_NT - 1) / PNG_NT + 3) & ~3);
  const int b = min(tid * piece, row_bytes), e = min(b + piece, row_bytes);
  const PieceMasks m = build_masks(row32, b, e - b);
  // threads whose whole piece continues a run: merged along the wave, the first of a chain emits the stretch
  const bool full = e > b && m.eq == m.valid;
  const unsigned long long fbal = __ballot(full);
  const bool chained = full && lane > 0 && ((fbal >> (lane - 1)) & 1ull);
  int chain_bytes = full && !chained ? min(b + ctz64(~(fbal >> lane)) * piece, row_bytes) - b : 0;
  // (a chain shorter than a match can be -- the row's last piece, one to three bytes, on its own -- is walked like any other piece)
  const bool walk = !full || (!chained && chain_bytes < 4);
  if (chain_bytes < 4) chain_bytes = 0;
  // pass 1: bits of this thread's tokens, Adler terms
  uint32_t nbits = 0, a_sum = 0, b_sum = 0;
  if (chain_bytes) {
    emit_long_run(chain_bytes, [&](int run) { unsigned long long bits; int n; match_code(run, bits, n); nbits += (uint32_t)n; });
  } else if (walk) {
    walk_piece(m,
               [&](int i, int n) {
                 const unsigned long long g = (m.ge >> i) & (n >= 64 ? ~0ull : ((1ull << n) - 1ull));
                 nbits += 8u * (uint32_t)n + (uint32_t)__popcll(g);
               },
               [&](int run) { unsigned long long bits; int n; match_code(run, bits, n); nbits += (uint32_t)n; });
  }
  // Adler terms a word at a time (b is a multiple of 4 for every thread that owns bytes; the bytes behind the row are zero and
  // pieces end on word boundaries or at the row's end): with d0..d3 the bytes at positions p..p+3, sum d = sad(word, 0),
  // sum (L - pos) d = (L - p) sum d - (d1 + 2 d2 + 3 d3), and d1 + 2 d2 + 3 d3 = sad(d1, d3) + 2 sad(d2, d3).
  for (int i = b >> 2; b < e && 4 * i < e; ++i) {
    const uint32_t v = row32[i];
    const uint32_t sd = __builtin_amdgcn_sad_u8(v, 0u, 0u);
    const uint32_t wsum = __builtin_amdgcn_sad_u8(v & 0xFF00FF00u, 0u, 0u) + 2u * __builtin_amdgcn_sad_u8(v & 0xFFFF0000u, 0u, 0u);
    a_sum += sd;
    b_sum += (uint32_t)(row_bytes - 4 * i) * sd - wsum;             // < 2^32: 64 * 255 * 16384
  }
  // bit offsets: prefix sum over the 256 threads (wave scans + four wave totals through LDS)
  const uint32_t incl = wave_incl_scan_u32(nbits, lane);
  if (lane == 63) s_wave[wave] = incl;
  uint32_t ra = a_sum % ADLER_MOD, rb = b_sum % ADLER_MOD;          // residues < 2^16: the wave sums fit 32 bits
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { ra += (uint32_t)__shfl_xor((int)ra, d, 64); rb += (uint32_t)__shfl_xor((int)rb, d, 64); }
  if (lane == 0) { s_ad[wave][0] = ra; s_ad[wave][1] = rb; }
  __syncthreads();
  uint32_t base = 0, total = 0;
#pragma unroll
  for (int w = 0; w < PNG_NT / 64; ++w) { const uint32_t t = s_wave[w]; if (w < wave) base += t; total += t; }
  uint32_t pos = 3u + base + incl - nbits;                          // behind the 3-bit block header
  // pass 2: emit
  unsigned long long acc = 0ull;
  int nacc = (int)(pos & 31u);
  uint32_t w = pos >> 5;
  auto put = [&](unsigned long long bits, int n) {
    acc |= bits << nacc;
    nacc += n;
    if (nacc >= 32) { atomicOr(&s_out[w], (uint32_t)acc); acc >>= 32; nacc -= 32; ++w; }
  };
  if (chain_bytes) {
    emit_long_run(chain_bytes, [&](int run) { unsigned long long bits; int n; match_code(run, bits, n); put(bits, n); });
  } else if (walk) {
    walk_piece(m,
               [&](int i, int n) {
                 for (int k = 0; k < n; k += 4) {            // four independent byte loads, then their codes
                   uint32_t v[4];
#pragma unroll
                   for (int u = 0; u < 4; ++u) v[u] = row8[b + i + min(k + u, n - 1)];
#pragma unroll
                   for (int u = 0; u < 4; ++u)
                     if (k + u < n) { uint32_t cb; int cn; lit_code(v[u], cb, cn); put((unsigned long long)cb, cn); }
                 }
               },
               [&](int run) { unsigned long long bits; int n; match_code(run, bits, n); put(bits, n); });
  }
  if (nacc) atomicOr(&s_out[w], (uint32_t)acc);
  __syncthreads();                                                  // every wave's codes are in the LDS image
  if (tid == 0) {
    atomicOr(&s_out[0], 2u);                               // BFINAL = 0, BTYPE = 01 (fixed Huffman): bits 0, 1, 0
    uint32_t end_bits = 3u + total;
    end_bits += 7u;                                        // end of block: seven zero bits (already there)
    end_bits += 3u;                                        // empty stored block: BFINAL = 0, BTYPE = 00 (zero bits) ...
    const uint32_t byte0 = (end_bits + 7u) >> 3;           // ... padded to a byte boundary, then LEN = 0, NLEN = 0xFFFF
    uint8_t* ob = reinterpret_cast<uint8_t*>(s_out);
    ob[byte0 + 2] = 0xFF; ob[byte0 + 3] = 0xFF;
    s_misc[0] = byte0 + 4u;
    sizes[r] = byte0 + 4u;
    uint32_t ta = 0, tb = 0;
#pragma unroll
    for (int k = 0; k < PNG_NT / 64; ++k) { ta += s_ad[k][0] % ADLER_MOD; tb += s_ad[k][1] % ADLER_MOD; }
    adler[2 * r] = ta % ADLER_MOD; adler[2 * r + 1] = tb % ADLER_MOD;
  }
  __syncthreads();
  const uint32_t n_out = s_misc[0];
  uint32_t* dst = reinterpret_cast<uint32_t*>(slots + (size_t)r * slot_stride);
  for (uint32_t i = tid; i < (n_out + 3u) / 4u; i += PNG_NT) dst[i] = s_out[i];
}

// grid = height + 1 waves.  Wave r < height copies row r's block to its place in the stream; wave `height` writes the frame.
__global__ __launch_bounds__(64) void png_assemble_kernel(const uint8_t* __restrict__ slots, int slot_stride, const uint32_t* __restrict__ sizes,
                                                          const uint32_t* __restrict__ adler, int height, int row_bytes,
                                                          uint8_t* __restrict__ stream, uint32_t stream_capacity,
                                                          uint32_t* __restrict__ stream_len) {
  const int r = blockIdx.x, lane = threadIdx.x;
  const int upto = r < height ? r : height;
  uint32_t off = 0;
  for (int j = lane; j < upto; j += 64) off += sizes[j];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) off += (uint32_t)__shfl_xor((int)off, d, 64);
  off += 2u;                                               // zlib header
  if (r < height) {
    const uint32_t n = sizes[r];
    if (off + n + 9u > stream_capacity) return;            // cannot happen with the capacity the host sizes; never write outside
    // The block lands at an arbitrary byte offset of the stream: a few head bytes up to the first aligned stream word, then
    // whole words -- each the funnel shift of two aligned words of the (16-byte aligned) slot, eight per lane in flight -- then
    // the tail bytes.  (A byte-per-lane loop is one memory round trip per 64 bytes: 200 us for a 1080p frame.)
    const uint8_t* src = slots + (size_t)r * slot_stride;
    uint8_t* dst = stream + off;
    const uint32_t head = min(n, (uint32_t)((4u - (uint32_t)(reinterpret_cast<uintptr_t>(dst) & 3u)) & 3u));
    if ((uint32_t)lane < head) dst[lane] = src[lane];
    const uint32_t n_words = (n - head) >> 2, sh = (head & 3u) * 8u;
    const uint32_t* s32 = reinterpret_cast<const uint32_t*>(src);
    uint32_t* d32 = reinterpret_cast<uint32_t*>(dst + head);
    for (uint32_t j0 = 0; j0 < n_words; j0 += 64 * 8) {
      uint32_t lo[8], hi[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const uint32_t j = j0 + u * 64 + lane, q = (head >> 2) + j;     // stream word j = slot bytes head + 4 j .. + 3
        lo[u] = j < n_words ? s32[q] : 0u;
        hi[u] = (j < n_words && sh) ? s32[q + 1] : 0u;                  // (the slot is padded: q + 1 stays inside it)
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const uint32_t j = j0 + u * 64 + lane;
        if (j < n_words) d32[j] = sh ? __funnelshift_r(lo[u], hi[u], sh) : lo[u];
      }
    }
    const uint32_t done = head + 4u * n_words;
    if (done + (uint32_t)lane < n) dst[done + lane] = src[done + lane];
    return;
  }
  // Adler-32 of all scanlines from the rows' terms: A = 1 + sum A_r, B = sum B_r + L H + L sum_j (H - 1 - j) A_j  (mod 65521)
  unsigned long long sa = 0ull, sb = 0ull, sw = 0ull;
  for (int j = lane; j < height; j += 64) {
    const unsigned long long aj = adler[2 * j], bj = adler[2 * j + 1];
    sa += aj; sb += bj; sw += (unsigned long long)(height - 1 - j) * aj;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    sa += ((unsigned long long)(uint32_t)__shfl_xor((int)(sa >> 32), d, 64) << 32) | (uint32_t)__shfl_xor((int)sa, d, 64);
    sb += ((unsigned long long)(uint32_t)__shfl_xor((int)(sb >> 32), d, 64) << 32) | (uint32_t)__shfl_xor((int)sb, d, 64);
    sw += ((unsigned long long)(uint32_t)__shfl_xor((int)(sw >> 32), d, 64) << 32) | (uint32_t)__shfl_xor((int)sw, d, 64);
  }
  if (lane != 0) return;
  const unsigned long long L = (unsigned long long)row_bytes % ADLER_MOD;
  const uint32_t A = (uint32_t)((1ull + sa) % ADLER_MOD);
  const uint32_t B = (uint32_t)((sb % ADLER_MOD + (L * ((unsigned long long)height % ADLER_MOD)) % ADLER_MOD + (L * (sw % ADLER_MOD)) % ADLER_MOD) % ADLER_MOD);
  if (off + 9u > stream_capacity) { stream_len[0] = 0u; return; }
  stream[0] = 0x78; stream[1] = 0x01;                      // zlib: deflate, 32 KB window, no dictionary, fastest
  uint8_t* t = stream + off;
  t[0] = 0x01; t[1] = 0x00; t[2] = 0x00; t[3] = 0xFF; t[4] = 0xFF;     // final block: stored, empty
  t[5] = (uint8_t)(B >> 8); t[6] = (uint8_t)B; t[7] = (uint8_t)(A >> 8); t[8] = (uint8_t)A;   // Adler-32, big endian: B << 16 | A
  stream_len[0] = off + 9u;
}

}  // namespace omfs

using namespace omfs;

extern "C" int omfs_prepare_target(const uint8_t* src, int channels, int src_width, int src_height, const uint8_t* mask,
                                   int width, int height, const float* bg_host, float* out_f32, uint8_t* out_u8, void* stream) {
  OMFS_REQUIRE(src && bg_host && (out_f32 || out_u8), "null pointer");
  OMFS_REQUIRE((channels == 1 || channels == 3 || channels == 4) && src_width > 0 && src_height > 0 && width > 0 && height > 0, "shape");
  OMFS_REQUIRE(width <= src_width * 64 && height <= src_height * 64, "upscaling beyond 64x is not an image-loading case");
  hipLaunchKernelGGL(prepare_target_kernel, dim3(cdiv(width * height, 256)), dim3(256), 0, (hipStream_t)stream, src, channels,
                     src_width, src_height, mask, width, height, bg_host[0], bg_host[1], bg_host[2], out_f32, out_u8);
  OMFS_CHECK_HIP(hipGetLastError());
  return OMFS_OK;
}

extern "C" int omfs_png_slot_stride(int width) {
  const int row_bytes = 1 + 3 * width;
  return ((row_bytes * 9 + 7) / 8 + 32 + 15) / 16 * 16;      // every byte a 9-bit literal, plus block header, end of block, flush marker
}

extern "C" int omfs_png_deflate(const uint8_t* rows, int width, int height, uint8_t* slots, uint32_t* sizes, uint32_t* adler,
                                uint8_t* stream, uint32_t stream_capacity, uint32_t* stream_len, void* stream_hip) {
  OMFS_REQUIRE(rows && slots && sizes && adler && stream && stream_len && width > 0 && height > 0, "args");
  const int row_bytes = 1 + 3 * width, stride = omfs_png_slot_stride(width);
  OMFS_REQUIRE((size_t)stream_capacity >= (size_t)height * stride + 16, "stream_capacity < height * omfs_png_slot_stride(width) + 16");
  const size_t lds = (size_t)((row_bytes + 8 + 15) / 16 * 16) + (size_t)stride;
  OMFS_REQUIRE(lds <= 64 * 1024, "scanline too long for the LDS image of its deflate block");
  hipStream_t s = (hipStream_t)stream_hip;
  OMFS_REQUIRE(row_bytes <= 64 * PNG_NT, "scanline longer than 16384 bytes");
  hipLaunchKernelGGL(png_deflate_rows_kernel, dim3(height), dim3(PNG_NT), lds, s, rows, row_bytes, height, slots, stride, sizes, adler);
  OMFS_CHECK_HIP(hipGetLastError());
  hipLaunchKernelGGL(png_assemble_kernel, dim3(height + 1), dim3(64), 0, s, (const uint8_t*)slots, stride, (const uint32_t*)sizes,
                     (const uint32_t*)adler, height, row_bytes, stream, stream_capacity, stream_len);
  OMFS_CHECK_HIP(hipGetLastError());
  return OMFS_OK;
}

// Host-side fetch of a device-deflated frame.  `dev` = [stream length: 4 bytes | 12 bytes pad | zlib stream] as one allocation (the
// layout engine/trainer.py gives omfs_png_deflate's outputs), `host` = pinned memory of the same size.  The length is only known
// on the device, so the copy is speculative: 16 + guess_bytes in one transfer, the remainder -- rarely -- in a second one.
// `copy_stream` is a stream of the CALLING thread; it waits for `ready_event` (recorded behind the deflate), and this call
// synchronises it (the one entry point that blocks: it exists so that an encoder thread spends its wait outside the host
// language's interpreter lock, in ONE foreign call).  Returns the stream length, or a negative error.
extern "C" long long omfs_png_fetch(void* host, const void* dev, size_t capacity_bytes, size_t guess_bytes, void* copy_stream,
                                    void* ready_event) {
  if (!host || !dev || capacity_bytes < 32) return omfs::set_error(OMFS_ERR_ARG, "omfs_png_fetch: bad arguments");
  hipStream_t s = (hipStream_t)copy_stream;
  if (ready_event) OMFS_CHECK_HIP(hipStreamWaitEvent(s, (hipEvent_t)ready_event, 0));
  const size_t first = (16 + guess_bytes) < capacity_bytes ? (16 + guess_bytes) : capacity_bytes;
  OMFS_CHECK_HIP(hipMemcpyAsync(host, dev, first, hipMemcpyDeviceToHost, s));
  OMFS_CHECK_HIP(hipStreamSynchronize(s));
  const uint32_t n = *reinterpret_cast<const uint32_t*>(host);
  if (n == 0u || 16 + (size_t)n > capacity_bytes) return omfs::set_error(OMFS_ERR_ARG, "omfs_png_fetch: impossible stream length %u", n);
  if (16 + (size_t)n > first) {
    OMFS_CHECK_HIP(hipMemcpyAsync((uint8_t*)host + first, (const uint8_t*)dev + first, 16 + (size_t)n - first, hipMemcpyDeviceToHost, s));
    OMFS_CHECK_HIP(hipStreamSynchronize(s));
  }
  return (long long)n;
}

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

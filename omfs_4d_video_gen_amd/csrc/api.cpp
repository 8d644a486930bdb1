// Error reporting and ABI version for libomfs_splat.so.
#include <cstdarg>
#include <cstdio>
#include "../../include/omfs_splat.h"

namespace omfs {
static thread_local char g_last_error[512] = "";
int set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
  va_end(ap);
  return code;
}
}  // namespace omfs

extern "C" int omfs_abi_version(void) { return OMFS_ABI_VERSION; }
extern "C" const char* omfs_last_error(void) { return omfs::g_last_error; }

// One training view in one call (see the header): the separate entry points in their order, stopping at the first error.
// ABI 7: the call is made of two halves a data-parallel host can put a collective between.
static int view_args_ok(const omfs_view_step* v) {
  return v && v->g && v->face_xf && v->cam && v->rb && v->gb && v->reg && (v->target || (v->target_rgb8 && v->target_scratch)) &&
         v->loss_out && v->loss_scratch;
}

extern "C" int omfs_view_forward_composite_bwd(const omfs_view_step* v, float* drgb_out, void* stream) {
  if (!view_args_ok(v)) return omfs::set_error(OMFS_ERR_ARG, "omfs_view_forward_composite_bwd: null pointer");
  int rc;
  if ((rc = omfs_project_fwd(v->g, v->face_xf, v->cam, v->rb, stream))) return rc;
  if ((rc = omfs_bin_sort(v->g, v->cam, v->rb, stream))) return rc;
  if ((rc = omfs_composite_fwd(v->cam, v->rb, stream))) return rc;
  const float* target = v->target;
  if (!target) {
    if ((rc = omfs_rgb8_to_image(v->target_rgb8, v->cam->width, v->cam->height, v->target_scratch, stream))) return rc;
    target = v->target_scratch;
  }
  if ((rc = omfs_loss_l1_ssim(v->rb->image, target, v->cam->width, v->cam->height, v->lambda_dssim, const_cast<float*>(v->gb->dimage),
                              v->loss_out, v->loss_scratch, stream)))
    return rc;
  if ((rc = omfs_composite_bwd(v->cam, v->rb, v->gb, stream))) return rc;
  if (drgb_out) return omfs_extract_drgb(v->rb, v->gb->dsplat, v->g->n, v->g->n_pad, drgb_out, stream);
  return 0;
}

extern "C" int omfs_view_project_bwd(const omfs_view_step* v, void* stream) {
  if (!view_args_ok(v)) return omfs::set_error(OMFS_ERR_ARG, "omfs_view_project_bwd: null pointer");
  return omfs_project_bwd(v->g, v->face_xf, v->cam, v->rb, v->gb, v->reg, stream);
}

extern "C" int omfs_view_forward_backward(const omfs_view_step* v, void* stream) {
  if (!view_args_ok(v)) return omfs::set_error(OMFS_ERR_ARG, "omfs_view_forward_backward: null pointer");
  const int rc = omfs_view_forward_composite_bwd(v, nullptr, stream);
  return rc ? rc : omfs_view_project_bwd(v, stream);
}
